#!/bin/bash
# what do the consumers cost the 2^22-point real step?  1 / 32 / 128 clients (n = 720), same stream
set -u
O=gpurun_out/r05ap; mkdir -p $O; rm -f $O/s.jsonl
K="python tools/kernel_times.py --fft 22 --real --batch 512 --steps 40 --ring-mib 2100 --mode 2"
for rep in 1 2; do
for c in 1 32 128; do
timeout 300 $K --clients $c --tag cfg5_c$c | tail -1 >> $O/s.jsonl
done
done
