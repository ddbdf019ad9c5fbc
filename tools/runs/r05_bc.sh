#!/bin/bash
# end-of-round stress: 3 x 200 fuzz cases (chain in), determinism soaks (IQ, real, hand-off), chain pipeline soaks
set -u
O=gpurun_out/r05bc; mkdir -p $O
for seed in 101 202 303; do
timeout 1200 python tools/fuzz_parity.py 200 $seed > $O/fuzz_$seed.log 2>&1; echo "fuzz $seed rc=$? $(tail -1 $O/fuzz_$seed.log | cut -c1-120)"
done
timeout 600 python tools/soak.py 60 > $O/soak_iq.log 2>&1; echo "soak iq rc=$? $(tail -1 $O/soak_iq.log)"
timeout 600 python tools/soak.py 60 real > $O/soak_real.log 2>&1; echo "soak real rc=$? $(tail -1 $O/soak_real.log)"
timeout 900 python tools/soak_handoff.py > $O/soak_handoff.log 2>&1; echo "soak handoff rc=$? $(tail -1 $O/soak_handoff.log | cut -c1-160)"
timeout 600 python tools/soak_post.py 64 256 60 4 2>&1 | tail -1
timeout 600 python tools/soak_post.py 300 64 40 3 2>&1 | tail -1
