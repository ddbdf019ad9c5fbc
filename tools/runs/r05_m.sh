#!/bin/bash
# CUs reserved for the side streams (PSDR_GRID_RESERVE, tuning build): what it costs the plain step, what it gives the post chain
set -u
O=gpurun_out/r05m; mkdir -p $O; : > $O/reserve.jsonl
K="python tools/kernel_times.py --batch 512 --steps 40 --mode 2"
run() { # tag reserve args...
  t=$1; r=$2; shift; shift
  PSDR_LIB=build/variants/libpsdr_tuning.so PSDR_GRID_RESERVE=$r timeout 300 $K "$@" --tag "$t" 2>>$O/err.log | tail -1 >> $O/reserve.jsonl
}
for rep in 1 2; do
for r in 0 8 16; do
run cfg2_plain_r$r $r --fft 20 --clients 16 --ring-mib 1100
run cfg2_post_r$r $r --fft 20 --clients 16 --ring-mib 1100 --post
done
for r in 0 8; do
run c256_plain_r$r $r --fft 20 --clients 256 --ring-mib 1100
run c256_post_r$r $r --fft 20 --clients 256 --ring-mib 1100 --post
run cfg5_plain_r$r $r --fft 22 --real --clients 128 --ring-mib 2100
done
done
cat $O/reserve.jsonl | cut -c1-330
