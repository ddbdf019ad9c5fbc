#!/bin/bash
# issued instructions per launch of the second passes: IQ 2^20 (1024-point rows), real 2^21 (1024-point rows + untangle), real 2^22 (2048)
set -u
R=$(pwd); O=$R/gpurun_out/r05bd; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for wl in iq real21 real22; do
  case $wl in
    iq) K2="python $R/tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 3 --ring-mib 1100";;
    real21) K2="python $R/tools/kernel_times.py --fft 21 --real --clients 64 --batch 512 --steps 3 --ring-mib 1100";;
    real22) K2="python $R/tools/kernel_times.py --fft 22 --real --clients 128 --batch 512 --steps 3 --ring-mib 2100";;
  esac
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_$wl -o p -- $K2 > $O/pmc_$wl.log 2>&1
  python $R/tools/pmc_generic_summary.py $O/pmc_$wl $O/inst_$wl.json > /dev/null
done
