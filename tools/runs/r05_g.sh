set -x
rm -f gpurun_out/r05_g.jsonl
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload cfg2 --no-extra --no-cpu-baseline --no-post-chain 2> gpurun_out/r05_g_$tag.err | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['path']['kernels']
    print(json.dumps({'v':'$tag','value':d['value'],'ms':d['ms_per_step'],'steps':d['steps'],'p1_us':k.get('fft_pass1',{}).get('device_clock_us_median'),'p2_us':k.get('fft_pass2',{}).get('device_clock_us_median'),'flow':(d['path'].get('one_launch') or {}).get('flow_control_waits')}))
except Exception as e:
    print(json.dumps({'v':'$tag','error':repr(e)}))
" >> gpurun_out/r05_g.jsonl; tail -2 gpurun_out/r05_g_$tag.err | cut -c1-200; }
run base PSDR_RING=0
run ring16 PSDR_RING=1
run ring16n112 PSDR_RING=1 PSDR_RING_P1_WGS=112
run ring16n120 PSDR_RING=1 PSDR_RING_P1_WGS=120
run ring32n120 PSDR_RING=1 PSDR_RING_P1_WGS=120 PSDR_RING_FRAMES=32
run ring16n88 PSDR_RING=1 PSDR_RING_P1_WGS=88
cat gpurun_out/r05_g.jsonl
