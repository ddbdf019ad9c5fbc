#!/bin/bash
# which part of the (new) post chain costs the step what: PSDR_PC_ABL bits 1 no moving averages, 2 no gain, 4 no peak kernels; PSDR_PC_RESERVE
set -u
O=gpurun_out/r05v; mkdir -p $O; : > $O/abl.jsonl
K="python tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 120 --ring-mib 1100 --mode 2"
run() { PSDR_LIB=build/variants/libpsdr_tuning.so PSDR_PC_ABL=$2 PSDR_PC_RESERVE=$3 timeout 300 $K $4 --tag "$1" 2>>$O/err.log | tail -1 >> $O/abl.jsonl; }
for rep in 1 2; do
run plain 0 8 ""
run post 0 8 --post
run post_r0 0 0 --post
run no_ma2 1 8 --post
run no_gain 2 8 --post
run no_peak 4 8 --post
run no_recurrences 3 8 --post
run only_gather_hist_out 7 8 --post
run only_gather_hist_out_r0 7 0 --post
done
cut -c1-200 $O/abl.jsonl
