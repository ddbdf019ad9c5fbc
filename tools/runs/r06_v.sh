#!/bin/bash
# the committed default bench line of the round's final build (profiles/r06_bench_default.json)
set -u
O=gpurun_out/r06v; mkdir -p $O
timeout 1500 python bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06v/bench_default.json'))
print('cfg2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_frac'], 'cfg3', d['cfg3']['roofline']['frac'], d['cfg3']['ms_per_step'], 'cfg5', d['cfg5_share']['roofline']['frac'], d['cfg5_share']['ms_per_step'])
print('post', d['post_chain']['over_plain'], 'c256', d['clients256']['value'], d['clients256']['ms_per_step'], d['clients256']['post_chain']['over_plain'])
print('scaling', {k:(v['ms_per_step'], v['frac_of_hbm_peak']) for k,v in d['real_input_client_scaling']['by_clients'].items()})
for key,w in (('16',d['with_fetch']),('256',d['clients256'].get('with_fetch'))):
    print(key, {k:(v.get('step_without_fetch_ms'),v.get('ms_per_step'),v.get('over_step_without_fetch'),v.get('d2h_GB_per_s_sustained')) for k,v in w.items() if isinstance(v,dict)}, w.get('error'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
