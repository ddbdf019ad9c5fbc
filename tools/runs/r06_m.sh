#!/bin/bash
# bench.py's own workloads, round 5's library (git a84e110) against this round's, same box, interleaved three times
# (this round's library here already carries the paired quartet records of the 2^22-point frames)
set -u
R=$(pwd); O=$R/gpurun_out/r06m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_quantiser_edges.py tests/test_gpu_truth_f64.py tests/test_gpu_configs_full.py tests/test_gpu_bench_shapes.py -m gpu -q -x -k "not gpus_2" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
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
cat $O/ab.jsonl | sort
tail -3 $O/ab.err
