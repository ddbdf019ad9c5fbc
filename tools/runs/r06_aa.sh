#!/bin/bash
# (1) the churn test of the two AGC forms (fixed: a paused client has nothing to read) + the other users of the post chain
# (2) kernel durations and timeline of the step with 256 mixed clients + post chain, both forms (rocprofv3 --kernel-trace --stats,
#     the torch-free driver)
set -u
R=$(pwd); O=$R/gpurun_out/r06aa; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_level2.py tests/test_gpu_abi.py tests/test_gpu_fuzz_slice.py -m gpu -q -x -k "post_chain or level2 or fetch or fuzz or pcm" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for form in 1 0; do
  PSDR_BENCH_AGC_FORM=$form rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$form -o p -- python $R/tools/kernel_times.py --fft 20 --clients 256 --mixed --batch 512 --steps 30 --post --ring-mib 1100 > $O/stats$form.log 2>&1
  cp $O/stats$form/p_kernel_stats.csv $O/c256_form${form}_kernel_stats.csv
  python $R/tools/trace_timeline.py $O/stats$form/p_kernel_trace.csv 2 > $O/c256_form${form}_timeline.txt 2>&1
  rm -rf $O/stats$form
  head -14 $O/c256_form${form}_kernel_stats.csv | cut -c1-150
done
