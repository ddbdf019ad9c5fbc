#!/bin/bash
# end-of-round stress on the final build: pipeline soaks of the post chain (16 / 64 / 256 clients), hand-off soaks of the fused
# real pass (2^21 / 2^22), IQ / real soaks, 600 fuzz cases
set -u
R=$(pwd); O=$R/gpurun_out/r06ap; mkdir -p $O
for c in 16 64 256; do timeout 600 python tools/soak_post.py $c 256 24 3 > $O/soak_post_$c.log 2>&1; echo "soak_post $c rc=$? $(tail -1 $O/soak_post_$c.log | cut -c1-160)"; done
timeout 900 python tools/soak_handoff.py 21 8 > $O/soak_handoff21.log 2>&1; echo "soak_handoff 21 rc=$? $(tail -1 $O/soak_handoff21.log | cut -c1-120)"
timeout 900 python tools/soak_handoff.py 22 5 > $O/soak_handoff22.log 2>&1; echo "soak_handoff 22 rc=$? $(tail -1 $O/soak_handoff22.log | cut -c1-120)"
timeout 900 python tools/soak.py > $O/soak.log 2>&1; echo "soak rc=$? $(tail -1 $O/soak.log | cut -c1-200)"
timeout 1500 python tools/fuzz_parity.py 600 > $O/fuzz.log 2>&1; echo "fuzz rc=$? $(tail -1 $O/fuzz.log | cut -c1-200)"
