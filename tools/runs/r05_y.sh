#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r05y; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_default -o p -- python $R/tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 12 --ring-mib 1100 --mode 0 --post > $O/trace_default.log 2>&1
PSDR_LIB=$R/build/variants/libpsdr_tuning.so PSDR_PC_STREAMS=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_t2 -o p -- python $R/tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 12 --ring-mib 1100 --mode 0 --post > $O/trace_t2.log 2>&1
cd $R
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 120 --ring-mib 1100 --mode 2 --clients 16 --post"
for rep in 1 2; do
timeout 300 $K --tag default | tail -1 >> $O/s.jsonl
PSDR_LIB=build/variants/libpsdr_tuning.so PSDR_PC_STREAMS=2 timeout 300 $K --tag tuning_streams2 | tail -1 >> $O/s.jsonl
PSDR_LIB=build/variants/libpsdr_tuning.so timeout 300 $K --tag tuning_plain | tail -1 >> $O/s.jsonl
done
