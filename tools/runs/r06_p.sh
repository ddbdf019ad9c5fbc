#!/bin/bash
# round 5's library against this round's (one-wave column tail again), bench.py workloads + the client-count points, same box, interleaved
set -u
R=$(pwd); O=$R/gpurun_out/r06p; mkdir -p $O
export PSDR_LIB_LENIENT=1
for rep in 1 2 3; do
  for w in cfg3 cfg5 cfg2; do
    for v in now r05; do
      PSDR_LIB=$R/build/variants/libpsdr_$v.so timeout 300 python bench.py --workload $w --no-extra --no-cpu-baseline --no-post-chain --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['path']['kernels']
print(json.dumps({'tag':'${w}_${v}','ms_per_step':d['ms_per_step'],'frac':d['roofline']['frac'],'p1':k.get('fft_pass1',{}).get('device_clock_us_median'),'p2':k.get('fft_pass2',{}).get('device_clock_us_median')}))"
    done
  done
  for v in now r05; do
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 20 --clients 256 --mixed --batch 512 --steps 10 --tag kt_cfg2c256_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 21 --real --clients 256 --mixed --batch 512 --steps 10 --tag kt_cfg3c256_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 21 --real --clients 1024 --mixed --batch 512 --steps 10 --tag kt_cfg3c1024_$v
  done
done > $O/ab.jsonl 2> $O/ab.err
python - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r06p/ab.jsonl'):
    try: r=json.loads(l)
    except Exception: continue
    if r['tag'].startswith('kt_'): d[r['tag']].append((r['us_per_frame_total'], r.get('fft_pass1_median'), r.get('fft_pass2_median')))
    else: d[r['tag']].append((r['ms_per_step'], r['p1'], r['p2']))
for k,v in sorted(d.items()): print(k, v)
PY
tail -3 $O/ab.err
