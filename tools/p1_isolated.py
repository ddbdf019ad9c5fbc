#!/usr/bin/env python
"""Pass-1 / pass-2 kernel times with and without consumer kernels on the side stream (how much of a
pass's time is contention with the previous batch's consumers).
  python tools/p1_isolated.py [--real] [--fft 21]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantomsdr_amd import AudioClient, Context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fft", type=int, default=21)
ap.add_argument("--real", action="store_true")
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
N, F = 1 << args.fft, args.batch
R = N // 2 if args.real else N
for levels, ncl in ((4, 0), (11, 0), (11, 16), (11, 64)):
    ctx = Context(N, args.real, levels, additional_size=360, audio_fft_size=360, input_format="s16", max_batch=F,
                  max_clients=max(ncl, 1))
    hb = ctx.half_frame_bytes()
    rng = np.random.default_rng(0)
    raw = rng.integers(-64, 64, size=(F + 1) * hb // 2, dtype=np.int16)
    d = ctx.dev_alloc(raw.nbytes)
    ctx.h2d(d, raw)
    cl = []
    for i in range(ncl):
        c = AudioClient(ctx)
        c.set_audio_demodulation(["USB", "LSB", "AM", "FM"][i % 4])
        m = int(rng.uniform(0.1 * R, 0.9 * R))
        c.set_audio_range(m, float(m), m + 89)
        cl.append(c)
    for i in range(3):
        ctx.process_batch(d, F)
        if ncl:
            ctx.demod_batch(i * F)
    ctx.synchronize()
    ctx.set_profiling(True)
    ctx.reset_kernel_stats()
    for i in range(10):
        ctx.process_batch(d, F)
        if ncl:
            ctx.demod_batch(i * F)
    ctx.synchronize()
    st = ctx.kernel_stats()
    print(json.dumps({"levels": levels, "clients": ncl, **{k: round(ms / n * 1e3, 1) for k, (ms, n) in st.items()}}))
    ctx.dev_free(d)
    ctx.close()
