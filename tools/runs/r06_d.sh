#!/bin/bash
# demodulation butterflies with literal roots (8 = 2.2.2, 9 = 3.3, 5, 10 = 2.5) against round 5's direct R-point DFTs: step and kernel
# times with 256 mixed clients on cfg2's stream and 1024 on cfg3's, same box, interleaved; then the parity subset
set -u
O=gpurun_out/r06d; mkdir -p $O
R=$(pwd)
for rep in 1 2 3; do
  for v in base oldbfly; do
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 20 --clients 256 --mixed --batch 512 --steps 10 --tag iq20c256_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 21 --real --clients 1024 --mixed --batch 512 --steps 10 --tag real21c1024_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 22 --real --clients 128 --mixed --batch 512 --steps 6 --ring-mib 1024 --tag real22c128_$v
  done
done > $O/ab.jsonl 2> $O/ab.err
for v in base oldbfly; do
  PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 21 --real --clients 1024 --mixed --batch 512 --steps 10 --mode 1 --tag ev_real21c1024_$v
done >> $O/ab.jsonl 2>> $O/ab.err
python - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r06d/ab.jsonl'):
    try: r=json.loads(l)
    except Exception: continue
    if r['tag'].startswith('ev_'): print(r)
    else: d[r['tag']].append((r['us_per_frame_total'], r.get('fft_pass1_median'), r.get('fft_pass2_median')))
for k,v in d.items(): print(k, v)
PY
tail -3 $O/ab.err
timeout 1200 python -m pytest tests/test_gpu_abi.py tests/test_gpu_state_freeze.py tests/test_gpu_fuzz_slice.py tests/test_gpu_truth_f64.py -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $O/pytest.log
