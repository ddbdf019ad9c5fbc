#!/bin/bash
# two chain streams (default now): parity subset, reserve on / off
set -u
R=$(pwd); O=$R/gpurun_out/r05x; mkdir -p $O; : > $O/s.jsonl
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_freeze.py tests/test_gpu_abi.py tests/test_gpu_level2.py -m gpu -q -x -k "post or freeze or abi or level2" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -2 $O/pytest.log
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 120 --ring-mib 1100 --mode 2"
run() { PSDR_LIB=$2 PSDR_PC_RESERVE=$3 timeout 300 $K --clients $4 $5 --tag "$1" 2>>$O/err.log | tail -1 >> $O/s.jsonl; }
T=build/variants/libpsdr_tuning.so
for rep in 1 2 3; do
run plain_c16 "" 8 16 ""
run post_c16 "" 8 16 --post
run post_c16_r0 $T 0 16 --post
run post_c16_r8 $T 8 16 --post
run plain_c256 "" 8 256 ""
run post_c256 "" 8 256 --post
run post_c256_r0 $T 0 256 --post
done
cut -c1-160 $O/s.jsonl
