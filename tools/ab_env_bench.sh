#!/bin/bash
# Runs ON THE GPU BOX: bench value of workloads under environment settings, interleaved.
#   tools/ab_env_bench.sh <tag> "<workloads>" "name:VAR=val VAR2=val" "name2:" ...
O=gpurun_out/$1; mkdir -p $O; WLS=$2; shift; shift
for rep in ${REPS:-1 2}; do
  for spec in "$@"; do
    tag="${spec%%:*}"; envs="${spec#*:}"
    for wl in $WLS; do
      env $envs python bench.py --workload $wl --no-extra --no-cpu-baseline --no-post-chain 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['path']['kernels']; print(json.dumps({'v':'$tag','wl':'$wl','value':d['value'],'ms':d['ms_per_step'],'frac':d['path']['frac_of_hbm_peak'],'p1_us':k['fft_pass1'].get('device_clock_us_median'),'p2_us':k['fft_pass2'].get('device_clock_us_median')}))" >> $O/bench.jsonl
    done
  done
done
cat $O/bench.jsonl
