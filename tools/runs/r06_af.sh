#!/bin/bash
# (1) end-of-round stress of the new post chain: pipeline soaks (16 / 64 / 256 clients, back-to-back batches against the drained
#     sequence), the fuzz (300 cases) (2) the chain's kernels alone on an idle chip (rocprofv3 --kernel-trace --stats): what
#     bounds k_pc_agc / k_pc_ma2 themselves
set -u
R=$(pwd); O=$R/gpurun_out/r06af; mkdir -p $O
for c in 16 64 256; do timeout 600 python tools/soak_post.py $c 256 24 3 > $O/soak_post_$c.log 2>&1; echo "soak_post $c rc=$? $(tail -1 $O/soak_post_$c.log | cut -c1-160)"; done
timeout 900 python tools/fuzz_parity.py 300 > $O/fuzz.log 2>&1; echo "fuzz rc=$? $(tail -1 $O/fuzz.log | cut -c1-200)"
cd /tmp; export TMPDIR=/tmp
for form in 1 0; do
  PSDR_BENCH_AGC_FORM=$form rocprofv3 --kernel-trace --stats --output-format csv -d $O/alone$form -o p -- python $R/tools/chain_alone.py 256 512 20 > $O/alone$form.log 2>&1
  grep "k_pc_\|k_demod" $O/alone$form/p_kernel_stats.csv | cut -c1-140 > $O/chain_alone_form${form}.csv
  rm -rf $O/alone$form
  cat $O/chain_alone_form${form}.csv
done
