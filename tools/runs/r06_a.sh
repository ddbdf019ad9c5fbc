#!/bin/bash
# round-6 baseline on this round's boxes: full GPU suite, default bench line (the tree as round 5 left it)
set -u
O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/pytest.log | tail -2
timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06a/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cfg3']['roofline']['frac'], d['cfg5_share']['roofline']['frac'], d['post_chain']['over_plain'], d['clients256']['post_chain']['over_plain'])
PY
