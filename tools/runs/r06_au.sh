#!/bin/bash
# HBM traffic of the post chain's kernels BESIDE the passes (the step with 256 clients + chain, tools/kernel_times.py --post), both
# forms: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (r06_ag.sh measured the chain alone on the chip)
set -u
R=$(pwd); O=$R/gpurun_out/r06au; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for form in 1 0; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    PSDR_BENCH_AGC_FORM=$form rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${form}_$ctr -o p -- python $R/tools/kernel_times.py --fft 20 --clients 256 --mixed --batch 512 --steps 6 --post --ring-mib 1100 > $O/pmc_${form}_$ctr.log 2>&1
    python $R/tools/pmc_generic_summary.py $O/pmc_${form}_$ctr $O/form${form}_$ctr.json > /dev/null 2>&1
    rm -rf $O/pmc_${form}_$ctr
  done
done
cd $R
python - <<'PY'
import json
out={}
for form in (1,0):
    f=json.load(open(f'gpurun_out/r06au/form{form}_FETCH_SIZE.json')); w=json.load(open(f'gpurun_out/r06au/form{form}_WRITE_SIZE.json'))
    ks={}
    for k in sorted(set(f)|set(w)):
        ks[k.replace('void ','')]={'FETCH_SIZE_KiB':round(f.get(k,{}).get('FETCH_SIZE',0),1),'WRITE_SIZE_KiB':round(w.get(k,{}).get('WRITE_SIZE',0),1)}
    pc={k:v for k,v in ks.items() if 'k_pc_' in k}
    out[f'form{form}']={'chain_kernels':pc,'other_kernels':{k:v for k,v in ks.items() if 'k_pc_' not in k},
        'chain_fetch_MB_as_reported':round(sum(v['FETCH_SIZE_KiB'] for v in pc.values())*1024/1e6,1),'chain_write_MB':round(sum(v['WRITE_SIZE_KiB'] for v in pc.values())*1024/1e6,1)}
    print(form, out[f'form{form}']['chain_fetch_MB_as_reported'], out[f'form{form}']['chain_write_MB'])
    for k,v in pc.items(): print('   ',k,v)
json.dump(out,open('gpurun_out/r06au/chain_traffic_beside.json','w'),indent=1)
PY
