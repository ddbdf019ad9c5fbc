set -x
rm -f gpurun_out/r05_h.jsonl
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload cfg2 --no-extra --no-cpu-baseline --no-post-chain 2> gpurun_out/r05_h_$tag.err | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['path']['kernels']
    print(json.dumps({'v':'$tag','value':d['value'],'ms':d['ms_per_step'],'p1_us':k.get('fft_pass1',{}).get('device_clock_us_median'),'p2_us':k.get('fft_pass2',{}).get('device_clock_us_median'),'flow':(d['path'].get('one_launch') or {}).get('flow_control_waits')}))
except Exception as e:
    print(json.dumps({'v':'$tag','error':repr(e)}))
" >> gpurun_out/r05_h.jsonl; tail -2 gpurun_out/r05_h_$tag.err | cut -c1-200; }
for rep in 1 2; do
run base PSDR_RING=0
run ring8n112 PSDR_RING=1 PSDR_RING_P1_WGS=112 PSDR_RING_FRAMES=8
run ring8n120 PSDR_RING=1 PSDR_RING_P1_WGS=120 PSDR_RING_FRAMES=8
run ring16n112 PSDR_RING=1 PSDR_RING_P1_WGS=112
run ring16n128 PSDR_RING=1 PSDR_RING_P1_WGS=128
run ring4n112 PSDR_RING=1 PSDR_RING_P1_WGS=112 PSDR_RING_FRAMES=4
done
cat gpurun_out/r05_h.jsonl
