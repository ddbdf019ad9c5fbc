#!/bin/bash
# (1) fetch / abi / parity subset on the shipped library (result sets of the demodulation double-buffered, fetch guards per buffer)
# (2) round 5's library (r05, git a84e110, PSDR_LIB_LENIENT) against this round's (now), same box, interleaved, the BASELINE shapes
# (3) the default bench line
set -u
R=$(pwd); O=$R/gpurun_out/r06l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_abi.py tests/test_gpu_parity.py tests/test_gpu_level2.py tests/test_gpu_state_freeze.py -m gpu -q -x --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -7 $O/pytest.log
export PSDR_LIB_LENIENT=1
for rep in 1 2 3; do
  for v in now r05; do
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 20 --clients 16 --batch 512 --steps 10 --tag cfg2_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 20 --clients 256 --mixed --batch 512 --steps 10 --tag cfg2c256_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 21 --real --clients 64 --mixed --batch 512 --steps 10 --tag cfg3_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 21 --real --clients 1024 --mixed --batch 512 --steps 10 --tag cfg3c1024_$v
    PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 22 --real --clients 128 --mixed --batch 512 --steps 6 --ring-mib 1024 --tag cfg5_$v
  done
done > $O/ab.jsonl 2> $O/ab.err
unset PSDR_LIB_LENIENT
python - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r06l/ab.jsonl'):
    try: r=json.loads(l)
    except Exception: continue
    d[r['tag']].append((r['us_per_frame_total'], r.get('fft_pass1_median'), r.get('fft_pass2_median')))
for k,v in sorted(d.items()): print(k, v)
PY
tail -3 $O/ab.err
timeout 1200 python bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06l/bench_default.json'))
print('cfg2', d['value'], d['ms_per_step'], d['roofline']['frac'], 'cfg3', d['cfg3']['roofline']['frac'], d['cfg3']['ms_per_step'], 'cfg5', d['cfg5_share']['roofline']['frac'], d['cfg5_share']['ms_per_step'])
print('post', d['post_chain']['over_plain'], 'c256', d['clients256']['value'], d['clients256']['ms_per_step'], d['clients256']['post_chain']['over_plain'])
print('scaling', {k:(v['ms_per_step'], v['frac_of_hbm_peak']) for k,v in d['real_input_client_scaling']['by_clients'].items()})
for key,w in (('16',d['with_fetch']),('256',d['clients256'].get('with_fetch'))):
    print(key, {k:(v.get('ms_per_step'),v.get('over_step_without_fetch'),v.get('d2h_GB_per_s_sustained')) for k,v in w.items() if isinstance(v,dict)}, w.get('error'))
PY
