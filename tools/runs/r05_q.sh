#!/bin/bash
# post chain, fourth cut (two-wave moving averages, LDS reads a block ahead): parity, step, timeline
set -u
R=$(pwd); O=$R/gpurun_out/r05q; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_freeze.py tests/test_gpu_abi.py tests/test_gpu_level2.py -m gpu -q -x -k "post or freeze or abi or level2" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest.log
: > $O/pc.jsonl
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 40 --ring-mib 1100 --mode 2"
for rep in 1 2; do
  for c in 16 256; do
    timeout 300 $K --clients $c --tag plain_c$c 2>>$O/err.log | tail -1 >> $O/pc.jsonl
    timeout 300 $K --clients $c --post --tag post_c$c 2>>$O/err.log | tail -1 >> $O/pc.jsonl
  done
done
cut -c1-120 $O/pc.jsonl
cd /tmp; export TMPDIR=/tmp
for c in 16 256; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_c$c -o p -- python $R/tools/kernel_times.py --fft 20 --clients $c --batch 512 --steps 12 --ring-mib 1100 --post --mode 0 > $O/trace_c$c.log 2>&1
done
