#!/bin/bash
# k_col_tail with several waves per LDS image (ct4) against one wave per work-group (demod2 = the same library otherwise):
# parity of every pyramid-carrying shape first, then step / timeline at 256 and 1024 clients on cfg3's stream, cfg2 / cfg5 shapes
set -u
R=$(pwd); O=$R/gpurun_out/r06h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_quantiser_edges.py tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py tests/test_gpu_group.py -m gpu -q -x --durations=6 -k "not two_gpus" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 $O/pytest.log
for rep in 1 2 3; do
  for v in ct4 demod2; do
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 21 --real --clients 256 --mixed --batch 512 --steps 10 --tag real21c256_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 21 --real --clients 1024 --mixed --batch 512 --steps 10 --tag real21c1024_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 21 --real --clients 64 --mixed --batch 512 --steps 10 --tag real21c64_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 20 --clients 256 --mixed --batch 512 --steps 10 --tag iq20c256_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 10 --tag iq20c16_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 22 --real --clients 128 --mixed --batch 512 --steps 6 --ring-mib 1024 --tag real22c128_$v
  done
done > $O/ab.jsonl 2> $O/ab.err
python - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r06h/ab.jsonl'):
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
