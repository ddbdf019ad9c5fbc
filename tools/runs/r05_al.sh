#!/bin/bash
# CUs left free of the passes (PSDR_PC_RESERVE 8 / 16 / 24) x recurrence waves owning their SIMD or not
set -u
R=$(pwd); O=$R/gpurun_out/r05al; mkdir -p $O; rm -f $O/s.jsonl
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 120 --ring-mib 1100 --mode 2"
T=build/variants/libpsdr_tuning.so; H=build/variants/libpsdr_tuning_hog.so
for rep in 1 2; do
for c in 16 64 256; do
timeout 300 $K --clients $c --tag plain_c$c | tail -1 >> $O/s.jsonl
for r in 8 16 24; do
PSDR_LIB=$T PSDR_PC_RESERVE=$r timeout 300 $K --clients $c --post --tag post_c${c}_r$r | tail -1 >> $O/s.jsonl
PSDR_LIB=$H PSDR_PC_RESERVE=$r timeout 300 $K --clients $c --post --tag post_c${c}_r${r}_own | tail -1 >> $O/s.jsonl
done
done
done
