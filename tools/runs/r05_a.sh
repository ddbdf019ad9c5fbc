set -x
export PSDR_LIB=$PWD/build/variants/libpsdr_tuning.so
tools/ab_env_bench.sh r05_split "cfg5 iq21" "base:" "m2_2048:PSDR_LOG2M2=11" > /dev/null 2>&1
tools/ab_env_bench.sh r05_yalias "cfg2" "base:" "alias16:PSDR_Y_ALIAS=16" "alias4:PSDR_Y_ALIAS=4" > /dev/null 2>&1
unset PSDR_LIB
python bench.py > gpurun_out/r05_base_bench.json 2> gpurun_out/r05_base_bench.err
cat gpurun_out/r05_split/bench.jsonl gpurun_out/r05_yalias/bench.jsonl
