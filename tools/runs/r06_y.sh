#!/bin/bash
# what does each kernel of the post chain cost the step with 256 clients?  tuning build, PSDR_PC_SKIP = mask of chain kernels NOT
# launched (wrong results: a timing bound): 1 gather, 2 moving averages, 4 history, 8 sub-block maxima, 16 prefix maxima, 32 w_t,
# 64 gain, 128 int16 output.  Same box, interleaved twice.
set -u
R=$(pwd); O=$R/gpurun_out/r06y; mkdir -p $O
for rep in 1 2; do
  for m in 0 255 253 191 189 56 129 4 199 63 192; do
    PSDR_PC_SKIP=$m PSDR_LIB=$R/build/variants/libpsdr_tuning.so timeout 300 python bench.py --workload clients256 --no-extra --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['post_chain']
print(json.dumps({'skip':$m,'rep':$rep,'plain_ms':d['ms_per_step'],'chain_ms':p['ms_per_step'],'chain50_ms':p['ms_per_step_50_step_repetitions'],'over_plain':p['over_plain']}))"
  done
done > $O/skip.jsonl 2> $O/skip.err
sort $O/skip.jsonl
tail -3 $O/skip.err
