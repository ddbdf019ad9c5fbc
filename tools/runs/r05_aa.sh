#!/bin/bash
set -u
# two-wave gain: parity subset, step at 16 / 256 clients, timeline at 256
R=$(pwd); O=$R/gpurun_out/r05aa; mkdir -p $O; rm -f $O/s.jsonl
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_freeze.py tests/test_gpu_abi.py tests/test_gpu_level2.py -m gpu -q -x -k "post or freeze or abi or level2" 2>&1 | tail -2
K="python tools/kernel_times.py --fft 20 --batch 512 --steps 120 --ring-mib 1100 --mode 2"
for rep in 1 2 3; do
timeout 300 $K --clients 16 --tag plain_c16 | tail -1 >> $O/s.jsonl
timeout 300 $K --clients 16 --post --tag post_c16_q46 | tail -1 >> $O/s.jsonl
PSDR_LIB=build/variants/libpsdr_tuning.so PSDR_PC_STREAMS=2 timeout 300 $K --clients 16 --post --tag post_c16_q45 | tail -1 >> $O/s.jsonl
timeout 300 $K --clients 256 --tag plain_c256 | tail -1 >> $O/s.jsonl
timeout 300 $K --clients 256 --post --tag post_c256_q46 | tail -1 >> $O/s.jsonl
done
cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_c256 -o p -- python $R/tools/kernel_times.py --fft 20 --clients 256 --batch 512 --steps 12 --ring-mib 1100 --post --mode 0 > $O/trace_c256.log 2>&1; cd $R
timeout 900 python bench.py --no-extra 2> $O/bench.err | tail -1 > $O/bench_cfg2.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05aa/bench_cfg2.json'))
print(d['value'], d['ms_per_step'], d['post_chain'])
PY
