#!/bin/bash
# which hardware queues does the chain get inside bench.py (torch's streams exist there), and do the passes start late?
set -u
R=$(pwd); O=$R/gpurun_out/r05au; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/bench256 -o p -- python $R/bench.py --workload clients256 --no-extra --no-cpu-baseline --steps 20 --warmup 5 > $O/bench256.log 2>&1
tail -1 $O/bench256.log | cut -c1-300
