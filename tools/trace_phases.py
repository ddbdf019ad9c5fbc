#!/usr/bin/env python
"""Phase timeline (shader clock cycles) of work-group 0 in the two FFT passes.
Needs a -DPSDR_TRACE_ON build: PSDR_LIB=build/variants/libpsdr_trace.so python tools/trace_phases.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantomsdr_amd import SpectrumEngine  # noqa: E402

N, F = 1 << 20, int(os.environ.get("TRACE_F", "16"))
eng = SpectrumEngine(35_000_000, N, False, input_format="s16", max_batch=F, max_clients=1, max_waterfall_clients=1)
hb = eng.ctx.half_frame_bytes()
raw = np.random.default_rng(0).integers(-64, 64, size=(F * 4 + 1) * hb // 2, dtype=np.int16)
eng.upload_ring(raw)
for i in range(4):
    eng.step((i % 4) * F, F, demod=False, waterfall=False)
eng.ctx.synchronize()
buf = (C.c_ulonglong * 4864)()
fn = eng.ctx.lib._handle and C.CDLL(os.environ.get("PSDR_LIB")).psdr_debug_trace
fn.argtypes = [C.c_void_p, C.c_void_p]
rc = fn(eng.ctx.h, buf)
assert rc == 0, rc
allv = np.array(buf, dtype=np.int64)
t = np.stack([allv[0:128].reshape(8, 16), allv[2432:2560].reshape(8, 16)])
wg = np.stack([allv[256:2304].reshape(256, 8), allv[2688:4736].reshape(256, 8)])
names = {0: "top", 1: "xpose-wr(+ld wait)", 2: "bar", 3: "rd+bar+prefetch", 4: "stage0", 5: "bar", 6: "rd+bar",
         7: "stage1", 8: "bar", 9: "rd+bar", 10: "last stage+stores", 11: "bar", 12: "epilogue", 13: "bar(end)"}
for p in range(2):
    print(f"== pass {p + 1} (cycles, iteration: deltas between marks)")
    for it in range(4):
        row = t[p, it]
        marks = [(k, row[k]) for k in range(14) if row[k] != 0]
        if row[14] and row[15] and row[10]:
            ns = (row[15] - row[14]) * 10.0
            print(f"   wall {ns:.0f} ns for {row[10] - row[0]} ticks -> {(row[10] - row[0]) / ns:.2f} ticks/ns")
        marks.sort(key=lambda x: x[1])
        out = []
        for (k0, c0), (k1, c1) in zip(marks[:-1], marks[1:]):
            out.append(f"{names.get(k1, k1)}={c1 - c0}")
        tot = marks[-1][1] - marks[0][1] if marks else 0
        print(f" it{it}: total={tot}  " + "  ".join(out))
for p in range(2):
    w = wg[p].astype(np.float64)
    t0 = w[:, 0].min()
    rel = (w - t0) / 100.0  # us
    rel[w == 0] = np.nan
    names2 = ["entry", "prologue", "it0", "it1", "it2", "it3", "it4", "exit"]
    print(f"== pass {p + 1}: work-group timeline, us after the first entry (min / median / max over 256 WGs)")
    for k in range(8):
        col = rel[:, k]
        if np.all(np.isnan(col)):
            continue
        print(f"   {names2[k]:9s} {np.nanmin(col):7.2f} {np.nanmedian(col):7.2f} {np.nanmax(col):7.2f}")
    # per XCD (wg % 8) exit medians
    print("   exit by XCD:", " ".join(f"{np.nanmedian(rel[x::8, 7]):.1f}" for x in range(8)))
# device-side boundary between the two passes of the same step (100 MHz wall clock, absolute):
# last work-group of pass 1 out -> first work-group of pass 2 in -> its first tile's loads have landed
e1, x1 = wg[0][:, 0].astype(np.float64), wg[0][:, 7].astype(np.float64)
e2, pr2 = wg[1][:, 0].astype(np.float64), wg[1][:, 1].astype(np.float64)
ok = (e1 > 0) & (x1 > 0) & (e2 > 0)
if ok.any():
    print("== boundary pass 1 -> pass 2 (us): median exit -> last exit %.1f | last exit -> first entry %.1f | first -> last entry %.1f | "
          "entry -> prologue done (median) %.1f" % ((x1[ok].max() - np.median(x1[ok])) / 100, (e2[ok].min() - x1[ok].max()) / 100,
                                                   (e2[ok].max() - e2[ok].min()) / 100, np.median(pr2[ok] - e2[ok]) / 100))
eng.close()
