set -x
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload cfg2 --no-extra --no-cpu-baseline --no-post-chain 2> gpurun_out/r05_d_$tag.err | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['path']['kernels']
    print(json.dumps({'v':'$tag','value':d['value'],'ms':d['ms_per_step'],'frac':d['path']['frac_of_hbm_peak'],'p1_us':k.get('fft_pass1',{}).get('device_clock_us_median'),'p2_us':k.get('fft_pass2',{}).get('device_clock_us_median'),'fused_us':k.get('fft_fused',{}).get('device_clock_us_median')}))
except Exception as e:
    print(json.dumps({'v':'$tag','error':repr(e)}))
" >> gpurun_out/r05_d.jsonl; tail -3 gpurun_out/r05_d_$tag.err | cut -c1-300; }
rm -f gpurun_out/r05_d.jsonl
for rep in 1 2; do
run base PSDR_RING=0
run ring16 PSDR_RING=1
run ring8 PSDR_RING=1 PSDR_RING_FRAMES=8
run ring32 PSDR_RING=1 PSDR_RING_FRAMES=32
run ring16n112 PSDR_RING=1 PSDR_RING_P1_WGS=112
run ring16n104 PSDR_RING=1 PSDR_RING_P1_WGS=104
done
cat gpurun_out/r05_d.jsonl
export PSDR_RING=1
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -x -q -k "bench_launch_256_frames_vs_oracle and cfg2" 2>&1 | tail -3
