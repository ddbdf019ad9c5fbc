#!/bin/bash
for rep in 1 2; do
for g in 256 248 240; do
  PSDR_P1_GRID=$g PSDR_P2_GRID=$g python tools/kernel_times.py --fft 20 --clients 16 --batch 256 --steps 20 --post --tag g$g | grep "^{" | cut -c1-400
done
done
