#!/bin/bash
# usage: tools/runs/gpu_retry.sh <log> <timeout_s> '<command>'   (HERE, not on the GPU box: retries while every GPU slot is busy)
LOG=$1; TMO=$2; shift; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $TMO -- "$@" > $LOG 2>&1
  rc=$?
  if ! grep -q "status=transient" $LOG; then exit $rc; fi
  sleep 90
done
