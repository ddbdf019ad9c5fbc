#!/bin/bash
# is the chain kernel bound by the latency of ONE chain or by throughput?  frames per chain K = 4 / 8 / 16 / 32 (PSDR_DEMOD_K), 256 and 1024
# clients on cfg3's stream: step (mode 2) and the kernel's own duration (mode 1)
set -u
O=gpurun_out/r06g; mkdir -p $O
for c in 256 1024; do
  for k in 4 8 16 32; do
    PSDR_DEMOD_K=$k python tools/kernel_times.py --fft 21 --real --clients $c --mixed --batch 512 --steps 10 --tag real21_c${c}_K$k
    PSDR_DEMOD_K=$k python tools/kernel_times.py --fft 21 --real --clients $c --mixed --batch 512 --steps 10 --mode 1 --tag ev_real21_c${c}_K$k
  done
done > $O/ab.jsonl 2> $O/ab.err
python - <<'PY'
import json
for l in open('gpurun_out/r06g/ab.jsonl'):
    try: r=json.loads(l)
    except Exception: continue
    print({k:v for k,v in r.items() if k in ('tag','us_per_frame_total','demod_idft','pyramid_tail','fft_pass1','fft_pass2','fft_pass1_median','fft_pass2_median')})
PY
tail -3 $O/ab.err
