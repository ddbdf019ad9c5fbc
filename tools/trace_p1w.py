#!/usr/bin/env python
"""Phase timeline (shader clock cycles) of wave 0 of work-group 0 in k_fft_pass1_w.
Needs a -DPSDR_TRACE_ON build: PSDR_LIB=build/variants/libpsdr_trace.so python tools/trace_p1w.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantomsdr_amd import SpectrumEngine  # noqa: E402

N, F = 1 << 20, int(os.environ.get("TRACE_F", "64"))
eng = SpectrumEngine(35_000_000, N, False, input_format="s16", max_batch=F, max_clients=1, max_waterfall_clients=1)
hb = eng.ctx.half_frame_bytes()
raw = np.random.default_rng(0).integers(-64, 64, size=(F * 4 + 1) * hb // 2, dtype=np.int16)
eng.upload_ring(raw)
for i in range(4):
    eng.step((i % 4) * F, F, demod=False, waterfall=False)
eng.ctx.synchronize()
buf = (C.c_ulonglong * 4864)()
fn = C.CDLL(os.environ.get("PSDR_LIB")).psdr_debug_trace
fn.argtypes = [C.c_void_p, C.c_void_p]
assert fn(eng.ctx.h, buf) == 0
allv = np.array(buf, dtype=np.int64)
t = allv[0:128].reshape(8, 16)
names = {1: "convert", 2: "stage0", 8: "alpha-wait", 9: "alpha-write+fetch", 3: "exch1", 4: "stage1", 5: "exch2", 6: "beta-wait", 7: "beta-read",
         10: "twiddle+stores"}
order = [0, 1, 2, 8, 9, 3, 4, 5, 6, 7, 10]
for it in range(8):
    row = t[it]
    if row[0] == 0:
        continue
    pts = [(k, row[k]) for k in order if row[k]]
    d = [f"{names[k1]}={c1 - c0}" for (k0, c0), (k1, c1) in zip(pts[:-1], pts[1:])]
    nxt = t[it + 1][0] - row[0] if it < 7 and t[it + 1][0] else 0
    print(f"it{it}: total={row[10] - row[0]} period={nxt}  " + "  ".join(d))
eng.close()
