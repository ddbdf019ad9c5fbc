#!/bin/bash
# default build after the policy change (waves owning their SIMD, reserve scaled with the work-groups): parity subset, step, bench
set -u
R=$(pwd); O=$R/gpurun_out/r05am; mkdir -p $O; rm -f $O/s.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_freeze.py tests/test_gpu_abi.py tests/test_gpu_level2.py tests/test_gpu_bench_shapes.py -m gpu -q -x -k "post or freeze or abi or level2 or bench_launch" 2>&1 | tail -2
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 120 --ring-mib 1100 --mode 2"
for rep in 1 2 3; do
for c in 16 64 256; do
timeout 300 $K --clients $c --tag plain_c$c | tail -1 >> $O/s.jsonl
timeout 300 $K --clients $c --post --tag post_c${c} | tail -1 >> $O/s.jsonl
done
done
timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json; echo "bench rc=$?"
