#!/bin/bash
# the final build of the round (int16 PCM option included): full GPU suite, smoke(), default bench line
set -u
O=gpurun_out/r06ai; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06ai/bench_default.json'))
print('cfg2', d['value'], d['ms_per_step'], d['roofline']['frac'], 'cfg3', d['cfg3']['roofline']['frac'], d['cfg3']['ms_per_step'], 'cfg5', d['cfg5_share']['roofline']['frac'], d['cfg5_share']['ms_per_step'])
print('post', d['post_chain']['over_plain'], 'c256', d['clients256']['value'], d['clients256']['ms_per_step'], d['clients256']['post_chain']['over_plain'])
print('scaling', {k:(v['ms_per_step'], v['frac_of_hbm_peak']) for k,v in d['real_input_client_scaling']['by_clients'].items()})
for key,w in (('16',d['with_fetch']),('256',d['clients256'].get('with_fetch'))):
    print(key, {k:(v.get('ms_per_step'),v.get('over_step_without_fetch'),v.get('d2h_GB_per_s_sustained')) for k,v in w.items() if isinstance(v,dict)}, w.get('error'))
PY
