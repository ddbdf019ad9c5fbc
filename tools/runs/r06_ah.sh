#!/bin/bash
# the post chain's PCM as int16 rows (PSDR_OPT_POST_CHAIN_PCM16): its test + the chain's other tests, then the served end with 256 and
# 16 clients (bench.py with_fetch: float audio / int32 PCM / int16 PCM)
set -u
R=$(pwd); O=$R/gpurun_out/r06ah; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py tests/test_gpu_level2.py -m gpu -q -x -k "post_chain or pcm or fetch or level2" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
for w in clients256 cfg2; do
  timeout 600 python bench.py --workload $w --no-extra --no-cpu-baseline --steps 40 --warmup 5 2> $O/bench_$w.err | tail -1 > $O/bench_$w.json
  python - <<PY
import json
d=json.load(open('$O/bench_$w.json'))
print('$w', d['ms_per_step'], 'chain', d['post_chain']['ms_per_step'], d['post_chain']['over_plain'])
for k,v in d['with_fetch'].items():
    if isinstance(v,dict): print('  ',k, v.get('ms_per_step'), v.get('over_step_without_fetch'), v.get('d2h_GB_per_s_sustained'), v.get('d2h_bytes_per_step'))
print(d['with_fetch'].get('error'))
PY
done
