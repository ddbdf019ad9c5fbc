#!/bin/bash
# pyramid tails enqueued BEHIND the demodulation (default now) against in front of it (PSDR_TAILS_FIRST=1, tuning build): parity subset
# on the shipped library first, then step times at 16 ... 1024 clients, same box, interleaved; timeline at 1024 clients
set -u
R=$(pwd); O=$R/gpurun_out/r06j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py tests/test_gpu_quantiser_edges.py tests/test_gpu_level2.py tests/test_gpu_command_timing.py tests/test_gpu_state_freeze.py tests/test_gpu_fuzz_slice.py -m gpu -q -x --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -9 $O/pytest.log
L=$R/build/variants/libpsdr_tuning.so
for rep in 1 2 3; do
  for tf in 0 1; do
    PSDR_TAILS_FIRST=$tf PSDR_LIB=$L python tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 10 --tag iq20c16_tf$tf
    PSDR_TAILS_FIRST=$tf PSDR_LIB=$L python tools/kernel_times.py --fft 20 --clients 256 --mixed --batch 512 --steps 10 --tag iq20c256_tf$tf
    PSDR_TAILS_FIRST=$tf PSDR_LIB=$L python tools/kernel_times.py --fft 21 --real --clients 64 --mixed --batch 512 --steps 10 --tag real21c64_tf$tf
    PSDR_TAILS_FIRST=$tf PSDR_LIB=$L python tools/kernel_times.py --fft 21 --real --clients 256 --mixed --batch 512 --steps 10 --tag real21c256_tf$tf
    PSDR_TAILS_FIRST=$tf PSDR_LIB=$L python tools/kernel_times.py --fft 21 --real --clients 1024 --mixed --batch 512 --steps 10 --tag real21c1024_tf$tf
    PSDR_TAILS_FIRST=$tf PSDR_LIB=$L python tools/kernel_times.py --fft 22 --real --clients 128 --mixed --batch 512 --steps 6 --ring-mib 1024 --tag real22c128_tf$tf
  done
done > $O/ab.jsonl 2> $O/ab.err
python - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r06j/ab.jsonl'):
    try: r=json.loads(l)
    except Exception: continue
    d[r['tag']].append((r['us_per_frame_total'], r.get('fft_pass1_median'), r.get('fft_pass2_median')))
for k,v in sorted(d.items()): print(k, v)
PY
tail -3 $O/ab.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o p -- python $R/tools/kernel_times.py --fft 21 --real --clients 1024 --mixed --batch 512 --steps 12 --mode 0 > $O/trace.log 2>&1
f=$(ls $O/trace/*/p_kernel_trace.csv $O/trace/p_kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/trace_timeline.py $f 2 > $O/timeline_c1024.txt; rm -rf $O/trace; cat $O/timeline_c1024.txt
