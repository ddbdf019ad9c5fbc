#!/bin/bash
# profiles of the FINAL build: cfg2 (bench line, kernel stats, PMC, TCC: tools/profile_round.sh) and the step with the post chain
# on at 16 and 256 clients (kernel stats + timeline), so that every r06_* file of the default paths is of the library that ships
set -u
R=$(pwd); O=$R/gpurun_out/r06an; mkdir -p $O
bash tools/profile_round.sh r06an cfg2 --no-cpu-baseline --no-post-chain > $O/profile_round.log 2>&1; tail -12 $O/profile_round.log
cd /tmp; export TMPDIR=/tmp
for c in 16 256; do
  M=""; [ $c = 256 ] && M="--mixed"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$c -o p -- python $R/tools/kernel_times.py --fft 20 --clients $c $M --batch 512 --steps 30 --post --ring-mib 1100 > $O/stats$c.log 2>&1
  cp $O/stats$c/p_kernel_stats.csv $O/r06_cfg2_post_chain_c${c}_kernel_stats.csv
  python $R/tools/trace_timeline.py $O/stats$c/p_kernel_trace.csv 2 > $O/r06_cfg2_post_chain_c${c}_timeline.txt 2>&1
  rm -rf $O/stats$c
  head -8 $O/r06_cfg2_post_chain_c${c}_kernel_stats.csv | cut -c1-140
done
