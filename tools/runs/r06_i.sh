#!/bin/bash
# (1) the group tests on the shipped library (overlapped exchange, migration with the fetched-set occupant check);
# (2) the one-launch transform pruned out of fft_pass.h: same-box A/B of the passes (prune) against the library before (ct4);
# (3) issued instructions of the demodulation chain kernel per transform (rocprofv3 --pmc, 1024 clients on cfg3's stream)
set -u
R=$(pwd); O=$R/gpurun_out/r06i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_abi.py tests/test_gpu_configs_full.py -m gpu -q -x --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -9 $O/pytest.log
for rep in 1 2 3; do
  for v in prune ct4; do
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 10 --tag iq20c16_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 21 --real --clients 64 --mixed --batch 512 --steps 10 --tag real21c64_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 22 --real --clients 128 --mixed --batch 512 --steps 6 --ring-mib 1024 --tag real22c128_$v
  done
done > $O/ab.jsonl 2> $O/ab.err
python - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r06i/ab.jsonl'):
    try: r=json.loads(l)
    except Exception: continue
    d[r['tag']].append((r['us_per_frame_total'], r.get('fft_pass1_median'), r.get('fft_pass2_median')))
for k,v in sorted(d.items()): print(k, v)
PY
cd /tmp; export TMPDIR=/tmp
K="python $R/tools/kernel_times.py --fft 21 --real --clients 1024 --mixed --batch 512 --steps 3 --mode 0"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $O/pmc -o p -- $K > $O/pmc.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc2 -o p -- $K > $O/pmc2.log 2>&1 </dev/null
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("gpurun_out/r06i/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        for key in ("k_demod_chain_fixed", "k_col_tail", "k_fft_pass2_real", "k_fft_pass1", "k_real_seam"):
            if key in k: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(acc):
    print(name, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(acc[name].items())}, "(millions per launch)")
PY
rm -rf $O/pmc $O/pmc2
