#!/bin/bash
# recurrence waves that own their SIMD (512 registers allocated: -DPSDR_PC_HOG) against the plain allocation
set -u
R=$(pwd); O=$R/gpurun_out/r05ak; mkdir -p $O; rm -f $O/s.jsonl
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 120 --ring-mib 1100 --mode 2"
T=build/variants/libpsdr_tuning.so; H=build/variants/libpsdr_tuning_hog.so
PSDR_LIB=$R/$H timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "post" 2>&1 | tail -2
for rep in 1 2 3; do
for c in 16 64 256; do
timeout 300 $K --clients $c --tag plain_c$c | tail -1 >> $O/s.jsonl
PSDR_LIB=$T timeout 300 $K --clients $c --post --tag post_c${c} | tail -1 >> $O/s.jsonl
PSDR_LIB=$H timeout 300 $K --clients $c --post --tag post_c${c}_own | tail -1 >> $O/s.jsonl
PSDR_LIB=$H PSDR_PC_LANES=64 timeout 300 $K --clients $c --post --tag post_c${c}_own_l64 | tail -1 >> $O/s.jsonl
done
PSDR_LIB=$H PSDR_PC_RESERVE=16 timeout 300 $K --clients 256 --post --tag post_c256_own_r16 | tail -1 >> $O/s.jsonl
done
cd /tmp; export TMPDIR=/tmp
PSDR_LIB=$R/$H timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c256 -o p -- python $R/tools/kernel_times.py --fft 20 --clients 256 --batch 512 --steps 40 --ring-mib 1100 --post --mode 0 > $O/trace.log 2>&1
head -6 $O/trace_c256/p_kernel_stats.csv | cut -c1-140
