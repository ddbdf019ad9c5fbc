#!/bin/bash
# 16 clients + post chain: four instead of eight CUs left free of the passes (the chain's two three-/four-wave work-groups need
# four; the grid then is not a multiple of 8: XCDs 0-3 carry one work-group more)?  tuning build, same box, interleaved
set -u
R=$(pwd); O=$R/gpurun_out/r06at; mkdir -p $O
for rep in 1 2 3; do
  for rs in 8 4; do
    PSDR_PC_RESERVE=$rs PSDR_PC_OWN=1 PSDR_LIB=$R/build/variants/libpsdr_tuning.so timeout 300 python bench.py --workload cfg2 --no-extra --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['post_chain']
print(json.dumps({'reserve':$rs,'rep':$rep,'plain_ms':d['ms_per_step'],'chain_ms':p['ms_per_step'],'over_plain':p['over_plain']}))"
  done
done > $O/ab.jsonl 2> $O/ab.err
sort $O/ab.jsonl; tail -2 $O/ab.err
