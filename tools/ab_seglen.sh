#!/bin/bash
# Runs ON THE GPU BOX: fused real pass 2 against the chain-segment length (cfg3 shape), interleaved
for rep in 1 2 3; do
  for sl in 4 8 16 32; do
    PSDR_SEG_LEN=$sl python tools/kernel_times.py --fft 21 --real --clients 64 --batch 256 --steps 10 --tag sl$sl | grep "^{" | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['tag'], j['fft_pass1'], j['fft_pass2'], j.get('real_seam'), j['us_per_frame_total'])"
  done
done
