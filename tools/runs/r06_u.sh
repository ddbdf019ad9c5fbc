#!/bin/bash
# cfg5's share: where do the 7 % of the step outside the two passes' own spans go?  kernel timeline of the plain step
set -u
R=$(pwd); O=$R/gpurun_out/r06u; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o p -- python $R/tools/kernel_times.py --fft 22 --real --clients 128 --mixed --batch 512 --steps 8 --ring-mib 1024 --mode 0 > $O/trace.log 2>&1
f=$(ls $O/trace/*/p_kernel_trace.csv $O/trace/p_kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/trace_timeline.py $f 2 > $O/timeline_cfg5.txt; rm -rf $O/trace; cat $O/timeline_cfg5.txt
tail -1 $O/trace.log | cut -c1-300
