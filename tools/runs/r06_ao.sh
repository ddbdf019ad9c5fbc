#!/bin/bash
# the floor of the post chain at 16 and 256 clients (tuning build): every chain kernel but the index kernel left out
# (PSDR_PC_SKIP=255), with and without the CUs left free for the chain (PSDR_PC_RESERVE), against the whole chain
set -u
R=$(pwd); O=$R/gpurun_out/r06ao; mkdir -p $O
for rep in 1 2; do
  for w in cfg2 clients256; do
    for v in "0:" "255:" "255:0" "0:8"; do
      m=${v%%:*}; rs=${v##*:}
      E="PSDR_PC_SKIP=$m PSDR_LIB=$R/build/variants/libpsdr_tuning.so"; [ -n "$rs" ] && E="$E PSDR_PC_RESERVE=$rs PSDR_PC_OWN=1"
      env $E timeout 300 python bench.py --workload $w --no-extra --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['post_chain']
print(json.dumps({'workload':'$w','skip':$m,'reserve':'$rs','rep':$rep,'plain_ms':d['ms_per_step'],'chain_ms':p['ms_per_step'],'over_plain':p['over_plain']}))"
    done
  done
done > $O/floor.jsonl 2> $O/floor.err
sort $O/floor.jsonl; tail -2 $O/floor.err
