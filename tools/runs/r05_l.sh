#!/bin/bash
# what does the post chain cost the step, and which part of it?  (tuning builds: PSDR_PC_ABL bits - 1 no k_pc_ma2, 2 no k_pc_gain,
# 4 no scan/want, 8 one stream, 16 normal-priority streams)
set -u
O=gpurun_out/r05l; mkdir -p $O; : > $O/pc_ab.jsonl
K="python tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 40 --ring-mib 1100 --mode 2"
run() { # tag lib abl post
  PSDR_LIB=build/variants/libpsdr_$2.so PSDR_PC_ABL=$3 timeout 300 $K $4 --tag "$1" 2>>$O/err.log | tail -1 >> $O/pc_ab.jsonl
}
for rep in 1 2; do
run plain tuning 0 ""
run post tuning 0 --post
run post_no_ma2 tuning 1 --post
run post_no_gain tuning 2 --post
run post_no_recurrences tuning 3 --post
run post_nothing_long tuning 7 --post
run post_normal_prio tuning 16 --post
run post_one_stream tuning 8 --post
run post_setprio0 tuning_noprio 0 --post
run post_setprio0_normal_prio tuning_noprio 16 --post
done
cat $O/pc_ab.jsonl | cut -c1-400
# instruction mix per launch of the two second passes (IQ 2^20 against real 2^22): what the untangle costs in issued instructions
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for wl in iq real; do
  if [ $wl = iq ]; then K2="python $R/tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 3 --ring-mib 1100"; else K2="python $R/tools/kernel_times.py --fft 22 --real --clients 128 --batch 512 --steps 3 --ring-mib 2100"; fi
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $R/$O/pmc_inst_$wl -o p -- $K2 > $R/$O/pmc_inst_$wl.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/$O/pmc_cyc_$wl -o p -- $K2 > $R/$O/pmc_cyc_$wl.log 2>&1
  python $R/tools/pmc_generic_summary.py $R/$O/pmc_inst_$wl $R/$O/inst_$wl.json > /dev/null
  python $R/tools/pmc_generic_summary.py $R/$O/pmc_cyc_$wl $R/$O/cyc_$wl.json > /dev/null
done
cd $R; cat $O/inst_iq.json $O/inst_real.json | head -80
