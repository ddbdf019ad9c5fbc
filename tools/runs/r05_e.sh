set -x
# the 1024 x 2048 split of 2^22-point real frames: the tests that touch that size first, then the cfg5 A/B
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_truth_f64.py -x -q -k "cfg5" 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_quantiser_edges.py tests/test_gpu_properties.py -x -q 2>&1 | tail -15
rm -f gpurun_out/r05_e.jsonl
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload cfg5 --no-extra --no-cpu-baseline --no-post-chain 2> gpurun_out/r05_e_$tag.err | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['path']['kernels']
    print(json.dumps({'v':'$tag','value':d['value'],'ms':d['ms_per_step'],'frac':d['path']['frac_of_hbm_peak'],'p1_us':k.get('fft_pass1',{}).get('device_clock_us_median'),'p2_us':k.get('fft_pass2',{}).get('device_clock_us_median')}))
except Exception as e:
    print(json.dumps({'v':'$tag','error':repr(e)}))
" >> gpurun_out/r05_e.jsonl; tail -2 gpurun_out/r05_e_$tag.err | cut -c1-300; }
for rep in 1 2 3; do
run split_2048x1024 PSDR_REAL_SPLIT=2048x1024
run split_1024x2048 PSDR_REAL_SPLIT=1024x2048
done
cat gpurun_out/r05_e.jsonl
