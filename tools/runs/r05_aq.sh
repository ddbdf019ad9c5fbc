#!/bin/bash
# prefix maxima behind the moving averages, w_t in front of the gain (default now) against all three peak kernels on the gain's stream
set -u
R=$(pwd); O=$R/gpurun_out/r05aq; mkdir -p $O; rm -f $O/s.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_freeze.py tests/test_gpu_level2.py -m gpu -q -x -k "post or freeze or level2" 2>&1 | tail -2
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 120 --ring-mib 1100 --mode 2"
T=build/variants/libpsdr_tuning.so
for rep in 1 2 3; do
for c in 16 64 256; do
timeout 300 $K --clients $c --tag plain_c$c | tail -1 >> $O/s.jsonl
PSDR_LIB=$T PSDR_PC_SPLIT_PEAK=1 timeout 300 $K --clients $c --post --tag post_c${c}_split | tail -1 >> $O/s.jsonl
PSDR_LIB=$T PSDR_PC_SPLIT_PEAK=0 timeout 300 $K --clients $c --post --tag post_c${c}_together | tail -1 >> $O/s.jsonl
done
done
