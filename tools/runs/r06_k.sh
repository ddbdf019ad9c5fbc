#!/bin/bash
# default bench line of the round's build (with_fetch included)
set -u
O=gpurun_out/r06k; mkdir -p $O
timeout 1200 python bench.py 2> $O/bench.err | tail -1 > $O/bench_default.json; echo "bench rc=$?"
tail -5 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06k/bench_default.json'))
print('cfg2', d['value'], d['ms_per_step'], d['roofline']['frac'], 'cfg3', d['cfg3']['roofline']['frac'], d['cfg3']['ms_per_step'], 'cfg5', d['cfg5_share']['roofline']['frac'], d['cfg5_share']['ms_per_step'])
print('post', d['post_chain']['over_plain'], 'c256', d['clients256']['value'], d['clients256']['ms_per_step'], d['clients256']['post_chain']['over_plain'])
print('scaling', {k:(v['ms_per_step'], v['frac_of_hbm_peak']) for k,v in d['real_input_client_scaling']['by_clients'].items()})
print('with_fetch', json.dumps(d['with_fetch'])[:1500])
print('c256 with_fetch', json.dumps(d['clients256'].get('with_fetch'))[:1500])
PY
