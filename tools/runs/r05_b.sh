set -x
export PSDR_LIB=$PWD/build/variants/libpsdr_tuning.so
python bench.py --workload cfg5 --no-extra --no-cpu-baseline --no-post-chain 2>&1 | tail -3 | cut -c1-600
tools/ab_env_bench.sh r05_split "cfg5 iq21" "base:" "m2_2048:PSDR_LOG2M2=11" > /dev/null 2>&1
tools/ab_env_bench.sh r05_yalias "cfg2" "base:" "alias16:PSDR_Y_ALIAS=16" "alias4:PSDR_Y_ALIAS=4" > /dev/null 2>&1
unset PSDR_LIB
cat gpurun_out/r05_split/bench.jsonl gpurun_out/r05_yalias/bench.jsonl
timeout 1200 python -m pytest tests/test_gpu_truth_f64.py tests/test_gpu_group.py -x -q 2>&1 | tail -15
