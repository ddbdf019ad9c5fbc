#!/usr/bin/env python
"""Determinism soak: the same 64-frame batch of 2^20-point IQ frames processed over and over
(with demodulation and waterfall running on the side stream) must give bit-identical spectra,
pyramids and audio every time - tile tickets, double-buffered result sets and the persistent
work-groups' LDS hazards all show up here if they are wrong.   tools/soak.py [iterations] [real]
With `real`: 2^21-point real frames (the fused second pass: chained tiles, carried rows in LDS, seam buffers
double-buffered against the side stream's seam kernel)."""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantomsdr_amd import SpectrumEngine  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
is_real = len(sys.argv) > 2 and sys.argv[2] == "real"
N, F = (1 << 21, 64) if is_real else (1 << 20, 64)
eng = SpectrumEngine(70_000_000 if is_real else 35_000_000, N, is_real, input_format="s16", max_batch=F, max_clients=16,
                     max_waterfall_clients=4)
hb = eng.ctx.half_frame_bytes()
rng = np.random.default_rng(7)
raw = rng.integers(-2000, 2000, size=(2 * F + 1) * hb // 2, dtype=np.int16)
eng.upload_ring(raw)
R = eng.params["fft_result_size"]
for i in range(16):
    m = int(rng.uniform(0.1 * R, 0.9 * R))
    mode = ("USB", "LSB", "AM", "FM")[i % 4]
    l, r = (m, m + 90) if mode == "USB" else (m - 90, m) if mode == "LSB" else (m - 90, m + 90)
    eng.add_audio_client(l, float(m), r, mode)
eng.add_waterfall_client(eng.params["downsample_levels"] - 1, 0, R >> (eng.params["downsample_levels"] - 1))
frames = [0, 17, 40, 63]
ref = None
bad = 0
for it in range(iters):
    # reset the carried demod state by re-adding nothing: instead always run the SAME two
    # batches (0 then 1) so that batch 1's audio depends on batch 0's tail identically
    eng.frame_num = 0
    eng.step(0, F)
    eng.step(F, F)
    eng.ctx.synchronize()
    sig = []
    for f in frames:
        sig.append(zlib.crc32(eng.ctx.read_spectrum(f).tobytes()))
        sig.append(zlib.crc32(eng.ctx.read_quantized(f).tobytes()))
    for c in eng.audio_clients[:4]:
        a, p, nan = c.read_audio(F)
        sig.append(zlib.crc32(a.tobytes()))
    if ref is None:
        ref = sig
    elif sig != ref:
        bad += 1
        print("iteration", it, "differs:", [i for i, (x, y) in enumerate(zip(sig, ref)) if x != y])
print(f"soak: {iters} iterations, {bad} mismatching")
eng.close()
sys.exit(1 if bad else 0)
