#!/bin/bash
# lanes of a recurrence wave in use (PSDR_PC_LANES, tuning build): 64 (one work-group per 64 slots), 32, 16
set -u
R=$(pwd); O=$R/gpurun_out/r05ai; mkdir -p $O; rm -f $O/s.jsonl
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 120 --ring-mib 1100 --mode 2"
T=build/variants/libpsdr_tuning.so
PSDR_LIB=$R/$T PSDR_PC_LANES=16 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_freeze.py -m gpu -q -x -k "post or freeze" 2>&1 | tail -2
for rep in 1 2 3; do
for c in 16 64 256; do
timeout 300 $K --clients $c --tag plain_c$c | tail -1 >> $O/s.jsonl
for l in 64 32 16; do
PSDR_LIB=$T PSDR_PC_LANES=$l timeout 300 $K --clients $c --post --tag post_c${c}_lanes$l | tail -1 >> $O/s.jsonl
done
done
done
cd /tmp; export TMPDIR=/tmp
PSDR_LIB=$R/$T PSDR_PC_LANES=16 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c256_l16 -o p -- python $R/tools/kernel_times.py --fft 20 --clients 256 --batch 512 --steps 40 --ring-mib 1100 --post --mode 0 > $O/trace.log 2>&1
head -8 $O/trace_c256_l16/p_kernel_stats.csv | cut -c1-140
