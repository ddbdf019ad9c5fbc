#!/bin/bash
# bench.py's workloads: round 5's library (git a84e110) against this round's (tails first again, paired quartet records, multi-wave column
# tail, the demodulation diet, double-buffered demodulation results), same box, interleaved three times
set -u
R=$(pwd); O=$R/gpurun_out/r06o; mkdir -p $O
export PSDR_LIB_LENIENT=1
for rep in 1 2 3; do
  for w in cfg3 cfg5 cfg2; do
    for v in now r05; do
      PSDR_LIB=$R/build/variants/libpsdr_$v.so timeout 300 python bench.py --workload $w --no-extra --no-cpu-baseline --no-post-chain --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['path']['kernels']
print(json.dumps({'tag':'${w}_${v}','ms_per_step':d['ms_per_step'],'frac':d['roofline']['frac'],'p1':k.get('fft_pass1',{}).get('device_clock_us_median'),'p2':k.get('fft_pass2',{}).get('device_clock_us_median')}))"
    done
  done
done > $O/ab.jsonl 2> $O/ab.err
sort $O/ab.jsonl
tail -3 $O/ab.err
