set -x
export PSDR_RING=1
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -x -q -k "bench_launch_256_frames_vs_oracle and cfg2" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_properties.py -x -q 2>&1 | tail -5
unset PSDR_RING
REPS="1 2" timeout 1500 tools/ab_env_bench.sh r05_ring "cfg2" "base:" "ring16:PSDR_RING=1" "ring8:PSDR_RING=1 PSDR_RING_FRAMES=8" "ring32:PSDR_RING=1 PSDR_RING_FRAMES=32" "ring16n104:PSDR_RING=1 PSDR_RING_P1_WGS=104" "ring16n88:PSDR_RING=1 PSDR_RING_P1_WGS=88" > /dev/null 2>&1
cat gpurun_out/r05_ring/bench.jsonl
