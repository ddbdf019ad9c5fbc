#!/bin/bash
# Runs ON THE GPU BOX: PCIe-inclusive rate of the whole Level-2 path from plain C (examples/stream_demo.c):
# cs16 samples from a file through pinned staging, the copy stream and the HBM ring, 2^20-point frames,
# 16 SSB audio clients + 4 waterfall clients, packets written to /dev/null.
#   tools/stream_rate.sh [frames per batch] [half-frames in the file]
set -e
B=${1:-64}; H=${2:-1025}
R=$(pwd)
gcc -O2 -I$R/include $R/examples/stream_demo.c -L$R/phantomsdr_amd -lpsdr_hip -lm -Wl,-rpath,$R/phantomsdr_amd -o /tmp/stream_demo
python - <<PY
import numpy as np
rng = np.random.default_rng(1)
h = (rng.standard_normal(($H, 1 << 20)) * 60).astype(np.int16)   # cs16: 2^19 complex samples per half-frame
h.tofile("/tmp/stream_in.cs16")
PY
A=""
for i in $(seq 0 15); do
  l=$((60000 + i * 61000)); A="$A --audio {\"cmd\":\"window\",\"l\":$l,\"r\":$((l + 89)),\"m\":$l.0} {\"cmd\":\"demodulation\",\"demodulation\":\"USB\"}"
done
W="--waterfall {\"cmd\":\"window\",\"l\":0,\"r\":1048576} --waterfall {\"cmd\":\"window\",\"l\":100000,\"r\":101024} --waterfall {\"cmd\":\"window\",\"l\":500000,\"r\":508192} --waterfall {\"cmd\":\"window\",\"l\":700000,\"r\":765536}"
cat /tmp/stream_in.cs16 > /dev/null   # page cache
for rep in 1 2 3; do
  s=$(date +%s.%N)
  /tmp/stream_demo 20 0 s16 35000000 12000 $B $A $W < /tmp/stream_in.cs16 > /dev/null 2> /tmp/stream_err.txt || { cat /tmp/stream_err.txt; exit 1; }
  e=$(date +%s.%N)
  python -c "
h=$H; t=$e-$s
print('stream_demo: %d frames of 2^20 points in %.3f s (incl. context creation): %.2f GS/s ingest, %.1f GB/s from the file' % (h-1, t, (h-1)*2**19/t/1e9, h*2**21/t/1e9))"
done
tail -1 /tmp/stream_err.txt
rm -f /tmp/stream_in.cs16
