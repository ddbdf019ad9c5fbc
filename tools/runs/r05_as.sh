#!/bin/bash
# the post chain at the audio rates of the reference's shipped configs (48 kHz: D = 128, 192 kHz: D = 512 - the generic moving-average kernels)
set -u
R=$(pwd); O=$R/gpurun_out/r05as; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for sps in 12000 48000 192000; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/a$sps -o p -- python $R/tools/kernel_times.py --fft 20 --clients 16 --batch 128 --steps 12 --ring-mib 600 --post --mode 0 --audio-sps $sps > $O/a$sps.log 2>&1
  tail -1 $O/a$sps.log | cut -c1-160
  grep "k_pc_\|demod" $O/a$sps/p_kernel_stats.csv | cut -d, -f1,2,4 | cut -c1-120
done
