#!/bin/bash
# the round's committed profile set (tools/profile_all.sh): per workload a bench line, rocprofv3 kernel stats, FETCH / WRITE / TCC passes;
# the post chain's kernel stats and timelines; the consumers alone against beside the passes
set -u
timeout 3000 bash tools/profile_all.sh r06 > gpurun_out/r06_profile_all.log 2>&1; echo "rc=$?"
ls gpurun_out/r06/profiles | head -40
tail -5 gpurun_out/r06_profile_all.log
