#!/bin/bash
# where does the step go with 1024 clients on cfg3's stream?  kernel timeline (rocprofv3 --kernel-trace) of the plain step
set -u
R=$(pwd); O=$R/gpurun_out/r06f; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in 1024 256; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_c$c -o p -- python $R/tools/kernel_times.py --fft 21 --real --clients $c --mixed --batch 512 --steps 12 --mode 0 > $O/trace_c$c.log 2>&1
  tail -1 $O/trace_c$c.log | cut -c1-200
  f=$(ls $O/trace_c$c/*/p_kernel_trace.csv $O/trace_c$c/p_kernel_trace.csv 2>/dev/null | head -1)
  python $R/tools/trace_timeline.py $f 3 > $O/timeline_c$c.txt
  rm -rf $O/trace_c$c
  cat $O/timeline_c$c.txt
done
