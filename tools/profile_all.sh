#!/bin/bash
# Runs ON THE GPU BOX: the round's committed profile set - per workload a bench line, rocprofv3 kernel stats and the PMC
# passes (tools/profile_round.sh), the post chain's kernel stats, the consumers alone vs beside the passes.
#   tools/profile_all.sh <tag>      -> gpurun_out/<tag>/profiles/*
TAG=${1:-r04}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O/profiles
cp profiles/traffic.json $O/profiles/traffic.json 2>/dev/null
for wl in cfg2 cfg3 cfg5 clients256; do
  timeout 900 tools/profile_round.sh $TAG $wl > $O/round_$wl.log 2>&1
done
cd /tmp; export TMPDIR=/tmp
for c in 16 256; do  # the step with the post chain on: kernel stats + a two-step timeline (queue ids included)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/post_c$c -o p -- python $R/tools/kernel_times.py --fft 20 --clients $c --batch 512 --steps 60 --ring-mib 1100 --post --mode 0 > $O/post_c$c.log 2>&1
  cp $O/post_c$c/p_kernel_stats.csv $O/profiles/${TAG}_cfg2_post_chain_c${c}_kernel_stats.csv 2>/dev/null
  python $R/tools/trace_timeline.py $O/post_c$c/p_kernel_trace.csv 2 > $O/profiles/${TAG}_cfg2_post_chain_c${c}_timeline.txt
done
cd $R
python tools/consumers_alone.py cfg2 cfg3 cfg5 clients256 --batch 512 > $O/profiles/${TAG}_consumers_alone_vs_beside.jsonl 2> $O/consumers.err
ls -la $O/profiles
