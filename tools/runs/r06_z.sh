#!/bin/bash
# the post chain's AGC as chunk maxima + ONE four-wave kernel (k_pc_cm / k_pc_cscan / k_pc_agc) against the five-kernel form:
# parity first (both forms against the oracle and against each other), then the step with 256 and 16 clients, same box, interleaved
set -u
R=$(pwd); O=$R/gpurun_out/r06z; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "post_chain" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.log
for rep in 1 2 3; do
  for form in 1 0; do
    for w in clients256 cfg2; do
      PSDR_BENCH_AGC_FORM=$form timeout 300 python bench.py --workload $w --no-extra --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['post_chain']
print(json.dumps({'workload':'$w','form':$form,'rep':$rep,'plain_ms':d['ms_per_step'],'chain_ms':p['ms_per_step'],'chain50_ms':p['ms_per_step_50_step_repetitions'],'over_plain':p['over_plain']}))"
    done
  done
done > $O/ab.jsonl 2> $O/ab.err
sort $O/ab.jsonl
tail -3 $O/ab.err
