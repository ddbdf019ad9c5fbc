#!/bin/bash
# two chain streams against three; plain trace for the dispatch gaps
set -u
R=$(pwd); O=$R/gpurun_out/r05w; mkdir -p $O; : > $O/s.jsonl
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 120 --ring-mib 1100 --mode 2"
run() { PSDR_LIB=build/variants/libpsdr_tuning.so PSDR_PC_STREAMS=$2 timeout 300 $K --clients $3 $4 --tag "$1" 2>>$O/err.log | tail -1 >> $O/s.jsonl; }
for rep in 1 2 3; do
run plain_c16 3 16 ""
run post3_c16 3 16 --post
run post2_c16 2 16 --post
run plain_c256 3 256 ""
run post3_c256 3 256 --post
run post2_c256 2 256 --post
done
cut -c1-160 $O/s.jsonl
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_plain -o p -- python $R/tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 12 --ring-mib 1100 --mode 0 > $O/trace_plain.log 2>&1
PSDR_LIB=$R/build/variants/libpsdr_tuning.so PSDR_PC_STREAMS=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_post2 -o p -- python $R/tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 12 --ring-mib 1100 --mode 0 --post > $O/trace_post2.log 2>&1
