#!/bin/bash
# k_pc_agc's three diets together (short division, one careful-vote per round, two alternating register sets of w: the kernel alone
# 1.94 -> 1.64 ms) against the library before them: does the STEP with 256 clients move?  five interleaved repetitions
set -u
R=$(pwd); O=$R/gpurun_out/r06al; mkdir -p $O
for rep in 1 2 3 4 5; do
  for v in now before; do
    E="PSDR_BENCH_AGC_FORM=1"; [ $v = before ] && E="$E PSDR_LIB=$R/build/variants/libpsdr_before.so PSDR_LIB_LENIENT=1"
    env $E timeout 300 python bench.py --workload clients256 --no-extra --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['post_chain']
print(json.dumps({'lib':'$v','rep':$rep,'plain_ms':d['ms_per_step'],'chain_ms':p['ms_per_step'],'chain50_ms':p['ms_per_step_50_step_repetitions'],'over_plain':p['over_plain']}))"
  done
done > $O/ab.jsonl 2> $O/ab.err
sort $O/ab.jsonl; tail -2 $O/ab.err
