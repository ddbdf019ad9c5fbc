#!/bin/bash
# Runs ON THE GPU BOX: per-kernel times of the fused real-input path (cfg3 shape) for the ablation
# builds of the library (tools/build_variants.py made them under build/variants/).
R=$(pwd)
for v in "$@"; do
  so=$R/build/variants/libpsdr_$v.so
  [ -f $so ] || continue
  PSDR_LIB=$so python tools/kernel_times.py --fft 21 --real --clients 16 --batch 256 --steps 10 --tag $v
done
