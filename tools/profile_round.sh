#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): bench line, rocprofv3 kernel stats and the two PMC
# passes for one workload; leaves everything under gpurun_out/<tag>/ and the summaries in
# gpurun_out/<tag>/profiles/ (copy those into profiles/ afterwards).
#   tools/profile_round.sh <tag> [workload] [extra bench args]
set -u
TAG=${1:-r01}; WL=${2:-cfg2}; shift; shift
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O/profiles
python bench.py --workload $WL --no-extra "$@" 2> $O/bench.err | tail -1 > $O/profiles/${TAG}_${WL}_bench.json
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --workload $WL --steps 30 --warmup 10 --no-cpu-baseline --no-post-chain --no-extra $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- $B > $O/stats.log 2>&1
# counter passes: the torch-free driver of the same library calls (rocprofv3's counter
# collection crashes on torch's own ring-generation kernels), same N / F / clients as the bench
# (512 frames per launch, as the bench: the fused real pass's hand-off plan starts there)
FPL=512
case $WL in
  cfg3) K="python $R/tools/kernel_times.py --fft 21 --real --clients 64 --batch $FPL --steps 3 --ring-mib 1100";;
  cfg5) K="python $R/tools/kernel_times.py --fft 22 --real --clients 128 --batch $FPL --steps 3 --ring-mib 2100";;
  clients256) K="python $R/tools/kernel_times.py --fft 20 --clients 256 --batch $FPL --steps 3 --ring-mib 1100";;
  *)    K="python $R/tools/kernel_times.py --fft 20 --clients 16 --batch $FPL --steps 3 --ring-mib 1100";;
esac
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- $K > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- $K > $O/pmc_write.log 2>&1
# L2 hit / miss of the same launches (is the inter-pass buffer served on chip?)
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -o p -- $K > $O/pmc_tcc.log 2>&1
cd $R
python tools/pmc_tcc_summary.py $O/pmc_tcc $O/profiles/${TAG}_${WL}_tcc.json
cp $O/stats/p_kernel_stats.csv $O/profiles/${TAG}_${WL}_kernel_stats.csv 2>/dev/null
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $WL $O/profiles/${TAG}_${WL}_pmc.json $O/profiles/traffic.json $FPL
cat $O/profiles/${TAG}_${WL}_bench.json
head -8 $O/profiles/${TAG}_${WL}_kernel_stats.csv | cut -c1-200
