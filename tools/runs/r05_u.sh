#!/bin/bash
# full GPU suite + default bench line after the post-chain rewrite
set -u
O=gpurun_out/r05u; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/pytest.log | tail -2
timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05u/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print('post', d['path']['post_chain'])
print('c256', {k:v for k,v in d['path']['clients256'].items() if k in ('value','ms_per_step','post_chain')})
PY
