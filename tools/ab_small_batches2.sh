#!/bin/bash
# Runs ON THE GPU BOX: cfg3's stream, the segment chooser of round 3 (PSDR_SEG_OLD) against the round-4 one and the hand-off plan
T=$PWD/build/variants/libpsdr_tuning.so
for F in $1; do for v in old new handoff; do
  case $v in old) E="PSDR_SEG_OLD=1 PSDR_SEG_HANDOFF=0";; new) E="PSDR_SEG_HANDOFF=0";; handoff) E="PSDR_SEG_HANDOFF_MIN=1";; esac
  env PSDR_LIB=$T $E python bench.py --workload ${WL:-cfg3} --batch $F --no-extra --no-cpu-baseline --no-post-chain 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'F':$F,'plan':'$v','value':round(d['value']/1e3,1),'ms':d['ms_per_step']}))"
done; done
