#!/usr/bin/env python
"""Level 1 of the drop-in (class hipFFT behind the reference's FFT interface): the per-frame cost of
load_*_input (two pinned f32 half-frames, H2D) + execute() (synchronous) + get_output_buffer() +
get_quantized_buffer() (D2H), i.e. everything the reference's fft_task does per frame with host buffers.
PCIe-inclusive by construction; bench.py's `value` never includes it.
    tools/level1_rate.py [log2 N] [real 0/1] [frames]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantomsdr_amd import core  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
is_real = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 200
N = 1 << m
levels = 11
cls = [getattr(core, n) for n in dir(core) if n.lower() == "hipfft"][0]
fft = cls(N, 1, levels, 0)
fft.set_output_additional_size(360)
fft.plan_r2c() if is_real else fft.plan_c2c(cls.FORWARD)
nfl = N // 2 if is_real else N            # floats per half-frame
bufs = [fft.malloc(nfl) for _ in range(3)]
rng = np.random.default_rng(0)
for b in bufs:
    b[:] = rng.standard_normal(nfl).astype(np.float32) * 0.01
load = fft.load_real_input if is_real else fft.load_complex_input
t_load = t_exec = t_out = 0.0
acc = 0.0
for i in range(frames + 5):
    if i == 5:
        t_load = t_exec = t_out = 0.0
    a1, a2 = bufs[i % 3], bufs[(i + 1) % 3]
    t0 = time.perf_counter()
    load(a1, a2)
    t1 = time.perf_counter()
    fft.execute()
    t2 = time.perf_counter()
    X = fft.get_output_buffer()
    q = fft.get_quantized_buffer()
    acc += float(X[1].real) + float(q[0])    # touch the host copies
    t3 = time.perf_counter()
    t_load += t1 - t0
    t_exec += t2 - t1
    t_out += t3 - t2
per = (t_load + t_exec + t_out) / frames
h2d = 2 * nfl * 4
d2h = X.nbytes + q.nbytes
print({"N": N, "real": is_real, "frames": frames, "us_per_frame": round(per * 1e6, 1),
       "load_us": round(t_load / frames * 1e6, 1), "execute_us": round(t_exec / frames * 1e6, 1),
       "get_buffers_us": round(t_out / frames * 1e6, 1), "MSamples_per_s": round(N / 2 / per / 1e6, 1),
       "h2d_bytes": h2d, "d2h_bytes": d2h, "pcie_GB_per_s": round((h2d + d2h) / per / 1e9, 1)})
fft.close()
