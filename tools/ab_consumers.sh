#!/bin/bash
# Runs ON THE GPU BOX: do the consumer kernels fit beside the real-input passes?  Library variants (tools/build_variants.py)
# interleaved on one box: bench value per workload.   usage: ab_consumers.sh <tag> "<variants>" "<workloads>" [bench args]
O=gpurun_out/${1:-ab_consumers}; mkdir -p $O
V=build/variants
VARS=${2:-"default p0 wpe6 wpe6p0"}; WLS=${3:-"cfg3 cfg5"}; shift; shift; shift
for rep in 1 2; do
  for v in $VARS; do
    lib=""; [ $v != default ] && lib=$PWD/$V/libpsdr_$v.so
    for wl in $WLS; do
      PSDR_LIB=$lib python bench.py --workload $wl --no-extra --no-cpu-baseline --no-post-chain "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'v':'$v','wl':'$wl','args':'$*','value':d['value'],'ms':d['ms_per_step'],'frac':d['path']['frac_of_hbm_peak'],'p1':d['roofline'].get('pass1_us'),'rf':d['roofline']['frac']}))" >> $O/bench.jsonl
    done
  done
done
cat $O/bench.jsonl
