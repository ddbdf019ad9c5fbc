#!/bin/bash
# Runs ON THE GPU BOX: A/B of library builds on the same box, interleaved repetitions (cfg2 shape)
R=$(pwd)
for rep in ${REPS:-1 2 3}; do
  for v in "$@"; do
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 20 --clients 16 --batch 256 --steps ${STEPS:-10} --tag $v
  done
done
