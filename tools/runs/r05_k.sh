#!/bin/bash
# round-5 final set after the knob prune: full GPU suite, default bench line, profile set of the four workloads
set -u
O=gpurun_out/r05k; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
tail -3 $O/pytest.log
timeout 600 python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json; echo "bench rc=$?" >> $O/rc.txt
cut -c1-600 $O/bench_default.json
for wl in cfg2 cfg3 cfg5 clients256; do
  timeout 900 bash tools/profile_round.sh r05 $wl > $O/profile_$wl.log 2>&1; echo "profile $wl rc=$?" >> $O/rc.txt
done
cat $O/rc.txt
ls gpurun_out/r05/profiles/
