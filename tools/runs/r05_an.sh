#!/bin/bash
# round-end state: full GPU suite, post-chain profiles of the final build, default bench line
set -u
R=$(pwd); O=$R/gpurun_out/r05an; mkdir -p $O/profiles
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/pytest.log | tail -2
cd /tmp; export TMPDIR=/tmp
for c in 16 256; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/post_c$c -o p -- python $R/tools/kernel_times.py --fft 20 --clients $c --batch 512 --steps 60 --ring-mib 1100 --post --mode 0 > $O/post_c$c.log 2>&1
  cp $O/post_c$c/p_kernel_stats.csv $O/profiles/r05_cfg2_post_chain_c${c}_kernel_stats.csv
  python $R/tools/trace_timeline.py $O/post_c$c/p_kernel_trace.csv 2 > $O/profiles/r05_cfg2_post_chain_c${c}_timeline.txt
done
cd $R
timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/profiles/r05_bench_default.json; echo "bench rc=$?"
