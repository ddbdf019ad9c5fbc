set -x
rm -f gpurun_out/r05_f.jsonl
run() { tag=$1; lib=$2; shift; shift; env PSDR_LIB=$lib "$@" timeout 300 python bench.py --workload cfg2 --no-extra --no-cpu-baseline --no-post-chain 2> gpurun_out/r05_f_$tag.err | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['path']['kernels']
    print(json.dumps({'v':'$tag','value':d['value'],'ms':d['ms_per_step'],'p1_us':k.get('fft_pass1',{}).get('device_clock_us_median'),'p2_us':k.get('fft_pass2',{}).get('device_clock_us_median'),'fused_us':k.get('fft_fused',{}).get('device_clock_us_median')}))
except Exception as e:
    print(json.dumps({'v':'$tag','error':repr(e)}))
" >> gpurun_out/r05_f.jsonl; tail -2 gpurun_out/r05_f_$tag.err | cut -c1-200; }
V=$PWD/build/variants
for rep in 1 2; do
run base $PWD/phantomsdr_amd/libpsdr_hip.so PSDR_RING=0
run ring $PWD/phantomsdr_amd/libpsdr_hip.so PSDR_RING=1
run ringst $V/libpsdr_ringst.so PSDR_RING=1
run ringld $V/libpsdr_ringld.so PSDR_RING=1
run ringstld $V/libpsdr_ringstld.so PSDR_RING=1
run ringnoflow $V/libpsdr_ringnoflow.so PSDR_RING=1
run ringall $V/libpsdr_ringall.so PSDR_RING=1
done
cat gpurun_out/r05_f.jsonl
