set -x
rm -f gpurun_out/r05_i.jsonl
run() { tag=$1; lib=$2; shift; shift; env PSDR_LIB=$lib "$@" timeout 300 python bench.py --workload cfg5 --no-extra --no-cpu-baseline --no-post-chain 2> gpurun_out/r05_i_$tag.err | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['path']['kernels']
    print(json.dumps({'v':'$tag','value':d['value'],'ms':d['ms_per_step'],'frac':d['path']['frac_of_hbm_peak'],'p1_us':k.get('fft_pass1',{}).get('device_clock_us_median'),'p2_us':k.get('fft_pass2',{}).get('device_clock_us_median')}))
except Exception as e:
    print(json.dumps({'v':'$tag','error':repr(e)}))
" >> gpurun_out/r05_i.jsonl; tail -2 gpurun_out/r05_i_$tag.err | cut -c1-200; }
P=$PWD/phantomsdr_amd/libpsdr_hip.so
for rep in 1 2; do
run old $P PSDR_REAL_SPLIT=2048x1024
run new $P
run new_nosplit $PWD/build/variants/libpsdr_nosplit.so
done
cat gpurun_out/r05_i.jsonl
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
