#!/bin/bash
# fetch ring of four host sets, the PCM's host lag 3: fetch tests, then bench.py's with_fetch blocks
set -u
O=gpurun_out/r06t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_abi.py tests/test_gpu_level2.py tests/test_gpu_group.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 1200 python bench.py --no-cpu-baseline 2> $O/bench.err | tail -1 > $O/bench_default.json; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06t/bench_default.json'))
print('cfg2', d['value'], d['ms_per_step'], d['roofline']['frac'], 'cfg3', d['cfg3']['roofline']['frac'], 'cfg5', d['cfg5_share']['roofline']['frac'])
for key,w in (('16',d['with_fetch']),('256',d['clients256'].get('with_fetch'))):
    print(key, {k:(v.get('step_without_fetch_ms'),v.get('ms_per_step'),v.get('over_step_without_fetch'),v.get('d2h_GB_per_s_sustained')) for k,v in w.items() if isinstance(v,dict)}, w.get('error'))
PY
