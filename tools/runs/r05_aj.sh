#!/bin/bash
set -u
O=gpurun_out/r05aj; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_freeze.py tests/test_gpu_abi.py tests/test_gpu_level2.py tests/test_gpu_bench_shapes.py -m gpu -q -x -k "post or freeze or abi or level2 or bench_launch" 2>&1 | tail -2
timeout 900 python tools/fuzz_parity.py 120 4242 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
