#!/bin/bash
# final check of the round: full GPU suite, smoke(), default bench line
set -u
O=gpurun_out/r05ah; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05ah/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['post_chain']['over_plain'], d['clients256']['post_chain']['over_plain'])
PY
