#!/bin/bash
# fuzz with the post chain in, undrained pipeline tests, determinism soaks on the round's kernels
set -u
O=gpurun_out/r05ae; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pipeline_matches" 2>&1 | tail -3
timeout 900 python tools/fuzz_parity.py 150 20250929 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz.log | cut -c1-300
timeout 900 python tools/fuzz_parity.py 100 777 > $O/fuzz2.log 2>&1; echo "fuzz2 rc=$?"; tail -2 $O/fuzz2.log | cut -c1-300
timeout 300 python tools/soak.py 30 > $O/soak_iq.log 2>&1; echo "soak iq rc=$?"; tail -1 $O/soak_iq.log
timeout 300 python tools/soak.py 30 real > $O/soak_real.log 2>&1; echo "soak real rc=$?"; tail -1 $O/soak_real.log
