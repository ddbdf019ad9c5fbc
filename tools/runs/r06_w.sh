#!/bin/bash
# end-of-round stress on the final build: 3 x 150 fuzz cases (post chain in), determinism soaks (IQ, real, hand-off), chain pipeline soaks
set -u
O=gpurun_out/r06w; mkdir -p $O
for seed in 11 22 33; do timeout 900 python tools/fuzz_parity.py 150 $seed > $O/fuzz_$seed.log 2>&1; echo "fuzz seed $seed rc=$? $(tail -1 $O/fuzz_$seed.log)"; done
timeout 600 python tools/soak.py > $O/soak.log 2>&1; echo "soak rc=$? $(tail -1 $O/soak.log | cut -c1-200)"
timeout 600 python tools/soak_handoff.py > $O/soak_handoff.log 2>&1; echo "soak_handoff rc=$? $(tail -1 $O/soak_handoff.log | cut -c1-200)"
timeout 600 python tools/soak_post.py > $O/soak_post.log 2>&1; echo "soak_post rc=$? $(tail -1 $O/soak_post.log | cut -c1-200)"
