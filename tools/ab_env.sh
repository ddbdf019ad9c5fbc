#!/bin/bash
# Runs ON THE GPU BOX: A/B of environment settings (tuning knobs of the library) on the same box, interleaved.
# usage: ab_env.sh "<kernel_times args>" "tag1:VAR=val VAR2=val" "tag2:" ...
KT="$1"; shift
for rep in ${REPS:-1 2 3}; do
  for spec in "$@"; do
    tag="${spec%%:*}"; envs="${spec#*:}"
    env $envs python tools/kernel_times.py $KT --tag "$tag"
  done
done
