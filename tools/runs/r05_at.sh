#!/bin/bash
# k_pc_mad (two-wave moving averages for any power-of-two delay): parity at 6 / 48 / 192 kHz, fuzz, kernel times at 48 and 192 kHz
set -u
R=$(pwd); O=$R/gpurun_out/r05at; mkdir -p $O
for k in 192000 48000 6000 44100 12000-8; do
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_post_chain_bit_exact and $k" 2>&1 | grep -E "passed|failed|fault|Error|assert" | head -3
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_state_freeze.py tests/test_gpu_level2.py -m gpu -q -x -k "post or freeze or level2" 2>&1 | tail -2
timeout 900 python tools/fuzz_parity.py 150 31337 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log | cut -c1-250
cd /tmp; export TMPDIR=/tmp
for sps in 48000 192000; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/a$sps -o p -- python $R/tools/kernel_times.py --fft 20 --clients 16 --batch 128 --steps 12 --ring-mib 600 --post --mode 0 --audio-sps $sps > $O/a$sps.log 2>&1
  python - $O/a$sps/p_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_pc_' in r['Name'] or 'demod' in r['Name'] or 'fft_pass' in r['Name']:
        print(f"  {r['Name'].split('(')[0][:48]:48s} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:10.1f} us")
PY
done
