#!/bin/bash
# chain streams chosen by measurement: first and later contexts of a process (tools/two_contexts.py), queue ids and launch gaps
set -u
R=$(pwd); O=$R/gpurun_out/r05aw; mkdir -p $O
for i in 1 2; do
PSDR_LIB=$R/build/variants/libpsdr_tuning.so PSDR_PC_VERBOSE=1 timeout 600 python tools/two_contexts.py 256 2>&1 | grep -E "psdr post chain|first context"
done
PSDR_LIB=$R/build/variants/libpsdr_tuning.so PSDR_PC_VERBOSE=1 timeout 600 python tools/two_contexts.py 16 2>&1 | grep -E "psdr post chain|first context"
cd /tmp; export TMPDIR=/tmp
PSDR_LIB=$R/build/variants/libpsdr_tuning.so PSDR_PC_VERBOSE=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/t -o p -- python $R/tools/two_contexts.py 256 > $O/t.log 2>&1
grep -E "psdr post chain|first context" $O/t.log
