#!/bin/bash
# moving averages reading the demodulator's rows themselves (k_pc_ma2 DIRECT: no gather for a work-group whose streams are whole)
# on top of the one-kernel AGC: parity first, then 256 / 16 clients against the library before (agc = commit 1b7..: one-kernel AGC
# with the gather) and against the five-kernel form, same box, interleaved
set -u
R=$(pwd); O=$R/gpurun_out/r06ab; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_level2.py tests/test_gpu_abi.py tests/test_gpu_fuzz_slice.py -m gpu -q -x -k "post_chain or level2 or fetch or fuzz or pcm" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
for rep in 1 2 3; do
  for v in direct agc five; do
    for w in clients256 cfg2; do
      case $v in
        direct) E="PSDR_BENCH_AGC_FORM=1";;
        agc)    E="PSDR_BENCH_AGC_FORM=1 PSDR_LIB=$R/build/variants/libpsdr_agc.so";;
        five)   E="PSDR_BENCH_AGC_FORM=0";;
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
