#!/bin/bash
# Runs ON THE GPU BOX: cfg3's stream at small batch sizes: uniform segments of every length against the hand-off plan.
T=$PWD/build/variants/libpsdr_tuning.so
for F in $1; do
  for SL in $2; do
    PSDR_LIB=$T PSDR_SEG_LEN=$SL python bench.py --workload cfg3 --batch $F --no-extra --no-cpu-baseline --no-post-chain 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'F':$F,'plan':'SL$SL','value':round(d['value']/1e3,1),'ms':d['ms_per_step']}))"
  done
  PSDR_LIB=$T PSDR_SEG_HANDOFF_MIN=1 python bench.py --workload cfg3 --batch $F --no-extra --no-cpu-baseline --no-post-chain 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'F':$F,'plan':'handoff','value':round(d['value']/1e3,1),'ms':d['ms_per_step']}))"
done
