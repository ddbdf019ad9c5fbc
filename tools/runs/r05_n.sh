#!/bin/bash
# kernel timeline of the step with the post chain on (16 and 256 clients): rocprofv3 --kernel-trace, start / end of every kernel
set -u
R=$(pwd); O=$R/gpurun_out/r05n; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in 16 256; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_c$c -o p -- python $R/tools/kernel_times.py --fft 20 --clients $c --batch 512 --steps 12 --ring-mib 1100 --post --mode 0 > $O/trace_c$c.log 2>&1
  tail -1 $O/trace_c$c.log | cut -c1-200
done
ls -la $O/trace_c16/* | head
