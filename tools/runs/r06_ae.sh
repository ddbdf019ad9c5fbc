#!/bin/bash
# demodulation: the scatter branch-free inside a round (bins without a place go to a dump line; one scalar branch per mode
# family, a scalar test per round) + slice positions of the first 256 bins kept per chain (HO = 4: 79 registers) - 'now' -
# against the library before ('before'): parity subset first, then step times with 256 mixed clients on cfg2's stream and 1024
# on cfg3's, same box, interleaved three times; issued instructions of the chain kernel per transform (rocprofv3 --pmc)
set -u
O=gpurun_out/r06ae; mkdir -p $O
R=$(pwd)
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py tests/test_gpu_state_freeze.py tests/test_gpu_fuzz_slice.py tests/test_gpu_truth_f64.py tests/test_gpu_configs_full.py -m gpu -q -x --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 $O/pytest.log
for rep in 1 2 3; do
  for v in now before; do
    L=$R/phantomsdr_amd/libpsdr_hip.so; [ $v = before ] && L=$R/build/variants/libpsdr_before.so
    PSDR_LIB=$L python tools/kernel_times.py --fft 20 --clients 256 --mixed --batch 512 --steps 10 --tag iq20c256_$v
    PSDR_LIB=$L python tools/kernel_times.py --fft 21 --real --clients 1024 --mixed --batch 512 --steps 10 --tag real21c1024_$v
    PSDR_LIB=$L python tools/kernel_times.py --fft 21 --real --clients 64 --mixed --batch 512 --steps 10 --tag real21c64_$v
  done
done > $O/ab.jsonl 2> $O/ab.err
python - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r06ae/ab.jsonl'):
    try: r=json.loads(l)
    except Exception: continue
    d[r['tag']].append((r['us_per_frame_total'], r.get('fft_pass1_median'), r.get('fft_pass2_median')))
for k,v in sorted(d.items()): print(k, v)
PY
tail -3 $O/ab.err
