#!/bin/bash
# chain streams by measurement (default) against creation order (PSDR_PC_PICK=0): first context (kernel_times) and third context (two_contexts)
set -u
R=$(pwd); O=$R/gpurun_out/r05ay; mkdir -p $O; rm -f $O/s.jsonl
T=$R/build/variants/libpsdr_tuning.so
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 120 --ring-mib 1100 --mode 2"
for rep in 1 2; do
for c in 16 256; do
PSDR_LIB=$T timeout 300 $K --clients $c --tag plain_c$c | tail -1 >> $O/s.jsonl
PSDR_LIB=$T PSDR_PC_PICK=1 timeout 300 $K --clients $c --post --tag post_c${c}_picked | tail -1 >> $O/s.jsonl
PSDR_LIB=$T PSDR_PC_PICK=0 timeout 300 $K --clients $c --post --tag post_c${c}_creation_order | tail -1 >> $O/s.jsonl
done
for c in 16 256; do
echo "picked:         $(PSDR_LIB=$T PSDR_PC_PICK=1 timeout 600 python tools/two_contexts.py $c 2>&1 | grep 'first context')"
echo "creation order: $(PSDR_LIB=$T PSDR_PC_PICK=0 timeout 600 python tools/two_contexts.py $c 2>&1 | grep 'first context')"
done
done
