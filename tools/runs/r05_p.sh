#!/bin/bash
# post chain, second cut (three chain streams, 256-row peak pieces, lane = slot gather / output): parity, step, timeline; PSDR_PC_RING variants
set -u
R=$(pwd); O=$R/gpurun_out/r05p; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_freeze.py tests/test_gpu_abi.py tests/test_gpu_level2.py -m gpu -q -x -k "post or freeze or abi or level2" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest.log
: > $O/pc.jsonl
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 40 --ring-mib 1100 --mode 2"
for rep in 1 2; do
  for c in 16 256; do
    timeout 300 $K --clients $c --tag plain_c$c 2>>$O/err.log | tail -1 >> $O/pc.jsonl
    timeout 300 $K --clients $c --post --tag post_c$c 2>>$O/err.log | tail -1 >> $O/pc.jsonl
    for v in ring8 ring10 ring12; do
      PSDR_LIB=build/variants/libpsdr_$v.so timeout 300 $K --clients $c --post --tag post_c${c}_$v 2>>$O/err.log | tail -1 >> $O/pc.jsonl
    done
  done
done
cut -c1-120 $O/pc.jsonl
cd /tmp; export TMPDIR=/tmp
for c in 16 256; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_c$c -o p -- python $R/tools/kernel_times.py --fft 20 --clients $c --batch 512 --steps 12 --ring-mib 1100 --post --mode 0 > $O/trace_c$c.log 2>&1
  PSDR_LIB=$R/build/variants/libpsdr_ring10.so timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_r10_c$c -o p -- python $R/tools/kernel_times.py --fft 20 --clients $c --batch 512 --steps 12 --ring-mib 1100 --post --mode 0 > $O/trace_r10_c$c.log 2>&1
done
