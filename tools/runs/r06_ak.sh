#!/bin/bash
# k_pc_agc with the short division (reciprocal + one residual correction, exhaustively equal to the IEEE division for the AGC's
# numerator: psdr_selftest_agc_division): the self-test and the chain's tests, the kernel alone, then 256 clients against the
# library before, same box, interleaved
set -u
R=$(pwd); O=$R/gpurun_out/r06ak; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py tests/test_gpu_level2.py -m gpu -q -x -k "post_chain or pcm or fetch or level2 or division" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.log
for rep in 1 2 3; do
  for v in now before; do
    E="PSDR_BENCH_AGC_FORM=1"; [ $v = before ] && E="$E PSDR_LIB=$R/build/variants/libpsdr_before.so PSDR_LIB_LENIENT=1"
    env $E timeout 300 python bench.py --workload clients256 --no-extra --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['post_chain']
print(json.dumps({'lib':'$v','rep':$rep,'plain_ms':d['ms_per_step'],'chain_ms':p['ms_per_step'],'chain50_ms':p['ms_per_step_50_step_repetitions'],'over_plain':p['over_plain']}))"
  done
done > $O/ab.jsonl 2> $O/ab.err
sort $O/ab.jsonl; tail -2 $O/ab.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/alone -o p -- python $R/tools/chain_alone.py 256 512 20 > $O/alone.log 2>&1
grep "k_pc_agc\|k_pc_ma2" $O/alone/p_kernel_stats.csv | cut -c1-140
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/tools/kernel_times.py --fft 20 --clients 256 --mixed --batch 512 --steps 30 --post --ring-mib 1100 > $O/stats.log 2>&1
grep "k_pc_agc\|k_pc_ma2" $O/stats/p_kernel_stats.csv | cut -c1-140
cp $O/stats/p_kernel_stats.csv $O/c256_kernel_stats.csv; rm -rf $O/stats $O/alone
