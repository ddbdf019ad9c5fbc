#!/bin/bash
R=$(pwd)
for rep in 1 2; do for v in fg1 fg4 fg8; do
  PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 20 --clients 256 --batch 256 --steps 30 --tag iq256_$v </dev/null | grep "^{" | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['tag'], j['fft_pass1'], j['fft_pass2'], j['demod_idft'], j['demod_ola'], j['us_per_frame_total'])"
  PSDR_LIB=$R/build/variants/libpsdr_$v.so python tools/kernel_times.py --fft 22 --real --clients 128 --batch 128 --steps 20 --tag cfg5_$v </dev/null | grep "^{" | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['tag'], j['fft_pass1'], j['fft_pass2'], j['demod_idft'], j['demod_ola'], j['us_per_frame_total'])"
done; done
