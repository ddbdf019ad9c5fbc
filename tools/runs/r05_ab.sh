#!/bin/bash
# the peak kernels in front of the gain recurrence (default) or behind the moving averages (PSDR_PC_STREAMS=1, tuning build)
set -u
R=$(pwd); O=$R/gpurun_out/r05ab; mkdir -p $O; rm -f $O/s.jsonl
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 120 --ring-mib 1100 --mode 2"
T=build/variants/libpsdr_tuning.so
for rep in 1 2 3; do
for c in 16 64 256; do
timeout 300 $K --clients $c --tag plain_c$c | tail -1 >> $O/s.jsonl
PSDR_LIB=$T timeout 300 $K --clients $c --post --tag post_c${c}_peak_sc | tail -1 >> $O/s.jsonl
PSDR_LIB=$T PSDR_PC_STREAMS=1 timeout 300 $K --clients $c --post --tag post_c${c}_peak_sm | tail -1 >> $O/s.jsonl
done
done
