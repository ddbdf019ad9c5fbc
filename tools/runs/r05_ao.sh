#!/bin/bash
# cfg5 share / cfg3: where does the step exceed the sum of the passes?  kernel timeline of the plain step
set -u
R=$(pwd); O=$R/gpurun_out/r05ao; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/cfg5 -o p -- python $R/tools/kernel_times.py --fft 22 --real --clients 128 --batch 512 --steps 14 --ring-mib 2100 --mode 0 > $O/cfg5.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/cfg3 -o p -- python $R/tools/kernel_times.py --fft 21 --real --clients 64 --batch 512 --steps 14 --ring-mib 1100 --mode 0 > $O/cfg3.log 2>&1
tail -1 $O/cfg5.log | cut -c1-200; tail -1 $O/cfg3.log | cut -c1-200
