#!/bin/bash
O=gpurun_out/r04_p0; mkdir -p $O
for rep in 1 2 3; do
  for v in default p0; do
    lib=""; [ $v != default ] && lib=$PWD/build/variants/libpsdr_$v.so
    for nc in 64 1024; do
      PSDR_LIB=$lib python tools/consumers_alone.py cfg3 --clients $nc --batch 512 --steps 8 --tag $v >> $O/cfg3.jsonl 2>> $O/err.log
    done
    PSDR_LIB=$lib python tools/consumers_alone.py cfg5 --batch 512 --steps 6 --tag $v >> $O/cfg5.jsonl 2>> $O/err.log
  done
done
