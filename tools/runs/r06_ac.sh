#!/bin/bash
# chunk maxima of the new samples from a third wave of the moving averages (k_pc_ma2 CMW; k_pc_cm only over the history chunks):
# parity first, then 256 / 16 clients against the library before (direct), same box, interleaved; kernel stats of the new form
set -u
R=$(pwd); O=$R/gpurun_out/r06ac; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_level2.py tests/test_gpu_abi.py tests/test_gpu_fuzz_slice.py -m gpu -q -x -k "post_chain or level2 or fetch or fuzz or pcm" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
for rep in 1 2 3; do
  for v in cmw direct; do
    for w in clients256 cfg2; do
      case $v in
        cmw)    E="PSDR_BENCH_AGC_FORM=1";;
        direct) E="PSDR_BENCH_AGC_FORM=1 PSDR_LIB=$R/build/variants/libpsdr_direct.so";;
      esac
      env $E timeout 300 python bench.py --workload $w --no-extra --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['post_chain']
print(json.dumps({'workload':'$w','lib':'$v','rep':$rep,'plain_ms':d['ms_per_step'],'chain_ms':p['ms_per_step'],'chain50_ms':p['ms_per_step_50_step_repetitions'],'over_plain':p['over_plain']}))"
    done
  done
done > $O/ab.jsonl 2> $O/ab.err
sort $O/ab.jsonl
tail -3 $O/ab.err
cd /tmp; export TMPDIR=/tmp
PSDR_BENCH_AGC_FORM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/tools/kernel_times.py --fft 20 --clients 256 --mixed --batch 512 --steps 30 --post --ring-mib 1100 > $O/stats.log 2>&1
cp $O/stats/p_kernel_stats.csv $O/c256_kernel_stats.csv
python $R/tools/trace_timeline.py $O/stats/p_kernel_trace.csv 2 > $O/c256_timeline.txt 2>&1
rm -rf $O/stats
head -14 $O/c256_kernel_stats.csv | cut -c1-150
