#!/bin/bash
# timing-only bounds (wrong results by construction) for VERDICT r5 items 1 and 2, same box, interleaved:
#   p1nox2   first pass without its second LDS exchange (what radix 32 x 32 could save at most)
#   r2nount  fused real second pass without the untangle arithmetic
#   r2rec16  2048-point rows: the two quartet records of a column in one 16-byte store (half the record stores)
set -u
O=gpurun_out/r06b; mkdir -p $O
R=$(pwd)
run() { # shape args..., variants
  :
}
for rep in 1 2 3; do
  for v in base p1nox2; do
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 10 --tag iq20_$v
  done
  for v in base r2nount p1nox2; do
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 21 --real --clients 64 --batch 512 --steps 10 --tag real21_$v
  done
  for v in base r2nount r2rec16; do
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 22 --real --clients 128 --batch 512 --steps 6 --ring-mib 1024 --tag real22_$v
  done
done > $O/ab.jsonl 2> $O/ab.err
python - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r06b/ab.jsonl'):
    try: r=json.loads(l)
    except Exception: continue
    d[r['tag']].append((r['us_per_frame_total'], r.get('fft_pass1_median'), r.get('fft_pass2_median')))
for k,v in d.items(): print(k, v)
PY
tail -3 $O/ab.err
