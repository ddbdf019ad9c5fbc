#!/bin/bash
# HBM traffic of the post chain's kernels per step, both forms (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, the
# chain alone on the chip: tools/chain_alone.py, 256 clients x 512 frames): the claim the round-6 form rests on
set -u
R=$(pwd); O=$R/gpurun_out/r06ag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for form in 1 0; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    PSDR_BENCH_AGC_FORM=$form rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${form}_$ctr -o p -- python $R/tools/chain_alone.py 256 512 6 > $O/pmc_${form}_$ctr.log 2>&1
    python $R/tools/pmc_generic_summary.py $O/pmc_${form}_$ctr $O/form${form}_$ctr.json > /dev/null 2>&1
    rm -rf $O/pmc_${form}_$ctr
  done
done
python - <<'PY'
import json
for form in (1,0):
    f=json.load(open(f'/root/repo/gpurun_out/r06ag/form{form}_FETCH_SIZE.json')) if False else json.load(open(f'gpurun_out/r06ag/form{form}_FETCH_SIZE.json')) if False else None
PY
cd $R
python - <<'PY'
import json
out={}
for form in (1,0):
    f=json.load(open(f'gpurun_out/r06ag/form{form}_FETCH_SIZE.json')); w=json.load(open(f'gpurun_out/r06ag/form{form}_WRITE_SIZE.json'))
    ks={}
    for k in sorted(set(f)|set(w)):
        if 'k_pc_' not in k: continue
        ks[k.replace('void ','')]={'FETCH_SIZE_KiB':round(f.get(k,{}).get('FETCH_SIZE',0),1),'WRITE_SIZE_KiB':round(w.get(k,{}).get('WRITE_SIZE',0),1)}
    tot_f=sum(v['FETCH_SIZE_KiB'] for v in ks.values()); tot_w=sum(v['WRITE_SIZE_KiB'] for v in ks.values())
    out[f'form{form}']={'kernels':ks,'fetch_MB_as_reported':round(tot_f*1024/1e6,1),'write_MB':round(tot_w*1024/1e6,1)}
    print(form, out[f'form{form}']['fetch_MB_as_reported'], out[f'form{form}']['write_MB'])
json.dump(out,open('gpurun_out/r06ag/chain_traffic.json','w'),indent=1)
print(json.dumps(out,indent=1)[:3000])
PY
