#!/bin/bash
set -u
O=gpurun_out/r05ad; mkdir -p $O
for k in "12000-8-5-248" "192000-128-5-248" "12000-7-6-252"; do
for i in 1 2 3; do
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "test_post_chain_bit_exact and $k" > $O/t_$k.$i.log 2>&1; echo "$k run $i rc=$?"
grep -i "fault\|abort\|passed\|failed\|Error" $O/t_$k.$i.log | grep -v "^  File" | head -5 | cut -c1-300
done
done
