#!/usr/bin/env python
"""Work-group timeline of the fused real-input pass 2 (segments finished per work-group, exit spread).
Needs a -DPSDR_TRACE_ON build: PSDR_LIB=build/variants/libpsdr_trace.so python tools/trace_real.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantomsdr_amd import SpectrumEngine  # noqa: E402

N, F = 1 << int(os.environ.get("TRACE_LOG2N", "21")), int(os.environ.get("TRACE_F", "256"))
eng = SpectrumEngine(70_000_000, N, True, input_format="s16", max_batch=F, max_clients=1, max_waterfall_clients=1)
hb = eng.ctx.half_frame_bytes()
raw = np.random.default_rng(0).integers(-64, 64, size=(F * 2 + 1) * hb // 2, dtype=np.int16)
eng.upload_ring(raw)
for i in range(4):
    eng.step((i % 2) * F, F, demod=False, waterfall=False)
eng.ctx.synchronize()
buf = (C.c_ulonglong * 4864)()
fn = C.CDLL(os.environ.get("PSDR_LIB")).psdr_debug_trace
fn.argtypes = [C.c_void_p, C.c_void_p]
assert fn(eng.ctx.h, buf) == 0
allv = np.array(buf, dtype=np.int64)
t = allv[2432:2560].reshape(8, 16)
names = {0: "top", 1: "xpose-wr(+ld wait)", 2: "bar", 3: "rd+bar", 4: "stage0", 5: "bar", 6: "rd+bar",
         7: "stage1", 8: "bar", 9: "rd+bar", 10: "last stage+untangle+stores", 11: "bar", 12: "octet loop", 13: "bar(end)"}
t1 = allv[0:128].reshape(8, 16)
names1 = {0: "top", 1: "xpose-wr(+ld wait)", 2: "bar", 3: "rd+bar+prefetch", 4: "stage0", 5: "bar", 6: "rd+bar",
          7: "stage1", 8: "bar", 9: "rd+bar", 10: "last stage+stores", 11: "bar", 12: "epilogue", 13: "bar(end)"}
print(f"pass 1 (paired), N = 2^{N.bit_length() - 1}, work-group 0, cycles between marks:")
for it in range(4):
    row = t1[it]
    marks = sorted([(k, row[k]) for k in range(14) if row[k] != 0], key=lambda x: x[1])
    out = [f"{names1.get(k1, k1)}={c1 - c0}" for (k0, c0), (k1, c1) in zip(marks[:-1], marks[1:])]
    print(f" it{it}: total={marks[-1][1] - marks[0][1] if marks else 0}  " + "  ".join(out))
print("pass 2 (real), work-group 0, cycles between marks:")
for it in range(1, 5):
    row = t[it]
    marks = sorted([(k, row[k]) for k in range(14) if row[k] != 0], key=lambda x: x[1])
    out = [f"{names.get(k1, k1)}={c1 - c0}" for (k0, c0), (k1, c1) in zip(marks[:-1], marks[1:])]
    print(f" it{it}: total={marks[-1][1] - marks[0][1] if marks else 0}  " + "  ".join(out))
w = allv[2688:4736].reshape(256, 8).astype(np.float64)
t0 = w[:, 0][w[:, 0] > 0].min()
rel = (w - t0) / 100.0
rel[w == 0] = np.nan
print("pass 2 (real): us after the first entry, min / median / max over 256 work-groups")
for k, name in enumerate(["entry", "seg1", "seg2", "seg3", "seg4", "seg5", "seg6", "exit"]):
    col = rel[:, k]
    if np.all(np.isnan(col)):
        continue
    print(f"   {name:6s} n={np.sum(~np.isnan(col)):3d} {np.nanmin(col):8.1f} {np.nanmedian(col):8.1f} {np.nanmax(col):8.1f}")
nseg = np.sum(~np.isnan(rel[:, 1:7]), axis=1)
print("   segments per work-group:", dict(zip(*np.unique(nseg, return_counts=True))))
print("   exit by XCD:", " ".join(f"{np.nanmedian(rel[x::8, 7]):.0f}" for x in range(8)))
w1 = allv[256:2304].reshape(256, 8).astype(np.float64)
rel1 = (w1 - w1[:, 0][w1[:, 0] > 0].min()) / 100.0
rel1[w1 == 0] = np.nan
print("pass 1: us after the first entry, min / median / max over 256 work-groups")
for k, name in enumerate(["entry", "prolog", "it0", "it1", "it2", "it3", "it4", "exit"]):
    col = rel1[:, k]
    if not np.all(np.isnan(col)):
        print(f"   {name:6s} n={np.sum(~np.isnan(col)):3d} {np.nanmin(col):8.1f} {np.nanmedian(col):8.1f} {np.nanmax(col):8.1f}")
print("   exit by XCD:", " ".join(f"{np.nanmedian(rel1[x::8, 7]):.0f}" for x in range(8)))
print("   exit - entry: min / median / max", np.nanmin(rel1[:, 7] - rel1[:, 0]), np.nanmedian(rel1[:, 7] - rel1[:, 0]), np.nanmax(rel1[:, 7] - rel1[:, 0]))
print("   exit percentiles 0/10/50/90/99/100:", np.nanpercentile(rel1[:, 7], [0, 10, 50, 90, 99, 100]).round(1))
eng.close()
