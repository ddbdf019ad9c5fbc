#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r05av; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/t -o p -- python $R/tools/two_contexts.py 256 > $O/t.log 2>&1
grep "first context" $O/t.log
