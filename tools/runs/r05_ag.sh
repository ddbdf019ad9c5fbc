#!/bin/bash
# the passes with a CU per XCD left free (what psdr_set_post_chain(1) does to their grids): the launch-shape and hand-off tests of the
# real and IQ plans on 248 work-groups (tuning build, PSDR_GRID_RESERVE=8)
set -u
O=gpurun_out/r05ag; mkdir -p $O
PSDR_LIB=$(pwd)/build/variants/libpsdr_tuning.so PSDR_GRID_RESERVE=8 timeout 1500 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_one_launch.py -m gpu -q -x -k "not gpus_2" > $O/pytest.log 2>&1; echo "rc=$?"
tail -3 $O/pytest.log
