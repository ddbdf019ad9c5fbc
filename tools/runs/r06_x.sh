#!/bin/bash
# octet records of the fused real pass paired like the quartets (a column's low and high record neighbours, their sums one 8-byte store):
# every 2^21 / 2^22-point test first, then bench.py's cfg3 / cfg5 with the library before (now) and after (pair8), same box, interleaved
set -u
R=$(pwd); O=$R/gpurun_out/r06x; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_quantiser_edges.py tests/test_gpu_truth_f64.py tests/test_gpu_configs_full.py tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q -x -k "not gpus_2" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.log
for rep in 1 2 3; do
  for w in cfg3 cfg5; do
    for v in pair8 now; do
      PSDR_LIB=$R/build/variants/libpsdr_$v.so timeout 300 python bench.py --workload $w --no-extra --no-cpu-baseline --no-post-chain --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['path']['kernels']
print(json.dumps({'tag':'${w}_${v}','ms_per_step':d['ms_per_step'],'frac':d['roofline']['frac'],'p1':k.get('fft_pass1',{}).get('device_clock_us_median'),'p2':k.get('fft_pass2',{}).get('device_clock_us_median')}))"
    done
  done
done > $O/ab.jsonl 2> $O/ab.err
sort $O/ab.jsonl
tail -3 $O/ab.err
timeout 900 python tools/soak_handoff.py 21 8 > $O/soak_handoff21.log 2>&1; echo "soak_handoff 21 rc=$? $(tail -1 $O/soak_handoff21.log | cut -c1-200)"
timeout 900 python tools/soak_handoff.py 22 5 > $O/soak_handoff22.log 2>&1; echo "soak_handoff 22 rc=$? $(tail -1 $O/soak_handoff22.log | cut -c1-200)"
