#!/bin/bash
# the round's new host-side pieces on the GPU: pipelined result fetch, the frame-ordered NaN guard (Inf / NaN bins and samples), the fuzz slice
set -u
O=gpurun_out/r06c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_abi.py tests/test_gpu_state_freeze.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $O/pytest.log
