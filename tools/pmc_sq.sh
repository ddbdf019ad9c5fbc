#!/bin/bash
# Runs ON THE GPU BOX: shader-side busy / wait / FIFO-full counters of the FFT passes (per launch, summed over the chip)
#   tools/pmc_sq.sh [iq|real]
R=$(pwd); W=${1:-iq}; O=$R/gpurun_out/pmc_sq_$W; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
if [ $W = real ]; then K="python $R/tools/kernel_times.py --fft 21 --real --clients 64 --batch 256 --steps 4"; else K="python $R/tools/kernel_times.py --fft 20 --clients 16 --batch 256 --steps 4"; fi
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/a -o p -- $K > $O/a.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/b -o p -- $K > $O/b.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --kernel-trace --output-format csv -d $O/c -o p -- $K > $O/c.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $O/d -o p -- $K > $O/d.log 2>&1 </dev/null
cd $R
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for part in "abcd":
    for fn in glob.glob("$O/%s/**/*counter_collection.csv" % part, recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "k_fft_pass" not in k:
                continue
            acc["pass1" if "pass1" in k else "pass2"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(acc):
    print("$W", name, {c: round(sum(v) / len(v) / 1e6, 1) for c, v in sorted(acc[name].items())}, "(millions per launch)")
PY
