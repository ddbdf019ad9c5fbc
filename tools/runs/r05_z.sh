#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r05z; mkdir -p $O; rm -f $O/s.jsonl
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 120 --ring-mib 1100 --mode 2"
for rep in 1 2 3; do
timeout 300 $K --clients 16 --tag plain_c16 | tail -1 >> $O/s.jsonl
timeout 300 $K --clients 16 --post --tag post_c16_q46 | tail -1 >> $O/s.jsonl
PSDR_LIB=build/variants/libpsdr_tuning.so PSDR_PC_STREAMS=2 timeout 300 $K --clients 16 --post --tag post_c16_q45 | tail -1 >> $O/s.jsonl
timeout 300 $K --clients 256 --tag plain_c256 | tail -1 >> $O/s.jsonl
timeout 300 $K --clients 256 --post --tag post_c256_q46 | tail -1 >> $O/s.jsonl
done
timeout 900 python bench.py --no-extra 2> $O/bench.err | tail -1 > $O/bench_cfg2.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05z/bench_cfg2.json'))
print(d['value'], d['ms_per_step'], d['post_chain'])
PY
