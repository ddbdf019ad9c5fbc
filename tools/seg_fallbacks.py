#!/usr/bin/env python
"""Fused real-input second pass with the hand-off plan: how many segments of a launch did not find their carried row in
memory in time and fell back to a seam (psdr_debug_seg_fallbacks).  usage: seg_fallbacks.py [log2N=21] [F=512] [launches=6]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantomsdr_amd import Context, _lib  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
F = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
N = 1 << log2n
ctx = Context(N, True, 12 if log2n == 22 else 11, input_format="s16", max_batch=F)
raw = np.random.default_rng(0).integers(-2000, 2000, size=(F + 1) * (N // 2), dtype=np.int16)
d = ctx.dev_alloc(raw.nbytes)
ctx.h2d(d, raw)
lib = _lib.load()
fn = lib.psdr_debug_seg_fallbacks
fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_void_p, C.c_uint]
out = []
for i in range(reps):
    ctx.process_batch(d, F)
    ns, fb = C.c_uint(0), C.c_uint(0)
    which = np.zeros(1 << 16, np.uint8)
    assert fn(ctx.h, C.byref(ns), C.byref(fb), which.ctypes.data, which.size) == 0
    out.append((ns.value, fb.value, which[:ns.value].reshape(-1, F).sum(axis=1).tolist() if ns.value else []))
print({"log2n": log2n, "frames": F, "segments_fallbacks_and_fallbacks_by_level_per_launch": out})
ctx.dev_free(d)
ctx.close()
