#!/bin/bash
# Runs ON THE GPU BOX: memory-side request counters (L2 <-> fabric) of the two second passes, per launch.
#   tools/pmc_ea.sh  -> gpurun_out/pmc_ea/{iq,real}_{a,b,c}/...  + a summary on stdout
R=$(pwd); O=$R/gpurun_out/pmc_ea; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
declare -A K
K[iq]="python $R/tools/kernel_times.py --fft 20 --clients 16 --batch 256 --steps 4"
K[real]="python $R/tools/kernel_times.py --fft 21 --real --clients 64 --batch 256 --steps 4"
for w in iq real; do
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --kernel-trace --output-format csv -d $O/${w}_a -o p -- ${K[$w]} > $O/${w}_a.log 2>&1 </dev/null
  timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum --kernel-trace --output-format csv -d $O/${w}_b -o p -- ${K[$w]} > $O/${w}_b.log 2>&1 </dev/null
  timeout 300 rocprofv3 --pmc TCC_BUSY_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_IB_STALL_sum --kernel-trace --output-format csv -d $O/${w}_c -o p -- ${K[$w]} > $O/${w}_c.log 2>&1 </dev/null
done
cd $R
python - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/pmc_ea"
for w in ("iq", "real"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for part in "abc":
        for fn in glob.glob(f"{O}/{w}_{part}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                k = r["Kernel_Name"]
                if "k_fft_pass" not in k:
                    continue
                name = "pass1" if "pass1" in k else "pass2"
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name in sorted(acc):
        print(w, name, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in sorted(acc[name].items())}, "(millions per launch)")
PY
