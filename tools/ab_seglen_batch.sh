#!/bin/bash
# Runs ON THE GPU BOX: cfg3's stream at batch sizes x uniform segment lengths (PSDR_SEG_LEN) - where the rounds of
# uniform segments stop being full.   tools/ab_seglen_batch.sh "<batches>" "<seg lens>"
for F in $1; do for SL in $2; do
  PSDR_SEG_LEN=$SL python bench.py --workload cfg3 --batch $F --no-extra --no-cpu-baseline --no-post-chain 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['path']['kernels']; print(json.dumps({'F':$F,'SL':$SL,'value':d['value'],'ms':d['ms_per_step'],'p2_us':k['fft_pass2'].get('device_clock_us_median')}))"
done; done
