#!/bin/bash
# where should a batch's consumers run?  bench.py's workloads with the tuning build: tails behind the demodulation (default) / in front
# (PSDR_TAILS_FIRST=1, round 5's order) / everything delayed by about one first pass (PSDR_SIDE_DELAY_US: beside the next SECOND pass)
set -u
R=$(pwd); O=$R/gpurun_out/r06n; mkdir -p $O
L=$R/build/variants/libpsdr_tuning.so
run() { # tag workload env...
  tag=$1; w=$2; shift; shift
  env "$@" PSDR_LIB=$L timeout 300 python bench.py --workload $w --no-extra --no-cpu-baseline --no-post-chain --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['path']['kernels']
print(json.dumps({'tag':'$tag','ms_per_step':d['ms_per_step'],'frac':d['roofline']['frac'],'p1':k.get('fft_pass1',{}).get('device_clock_us_median'),'p2':k.get('fft_pass2',{}).get('device_clock_us_median')}))"
}
for rep in 1 2 3; do
  run cfg3_deferred cfg3 PSDR_X=0
  run cfg3_tailsfirst cfg3 PSDR_TAILS_FIRST=1
  run cfg3_delay900 cfg3 PSDR_SIDE_DELAY_US=900
  run cfg3_delay600 cfg3 PSDR_SIDE_DELAY_US=600
  run cfg3_tf_delay500 cfg3 PSDR_TAILS_FIRST=1 PSDR_SIDE_DELAY_US=500
  run cfg2_deferred cfg2 PSDR_X=0
  run cfg2_tailsfirst cfg2 PSDR_TAILS_FIRST=1
  run cfg2_delay900 cfg2 PSDR_SIDE_DELAY_US=900
  run cfg5_deferred cfg5 PSDR_X=0
  run cfg5_tailsfirst cfg5 PSDR_TAILS_FIRST=1
  run cfg5_delay1800 cfg5 PSDR_SIDE_DELAY_US=1800
done > $O/ab.jsonl 2> $O/ab.err
sort $O/ab.jsonl
tail -3 $O/ab.err
