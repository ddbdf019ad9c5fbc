#!/usr/bin/env python
"""Soak of the in-launch hand-off of the fused real-input second pass (fft_pass.h): many 512-frame launches at random
offsets of a long raw stream, the int8 pyramid of EVERY frame against the same frames processed with whole-frame
segments (PSDR_SEG_LEN = G: no hand-off).  A stale carried row shows in the mirror-side octets of a segment's first tile.
usage: soak_handoff.py [log2N=21] [launches=30]"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantomsdr_amd import Context, _lib  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N, F, NH = 1 << log2n, 512, 1024  # NH half-frames of stream
G = (N // 2 // 1024) // 16
rng = np.random.default_rng(5)
raw = rng.integers(-3000, 3000, size=(NH + 1) * (N // 2), dtype=np.int16)


def digests(seg_len, offsets, want_stats=False):
    if seg_len:
        os.environ["PSDR_SEG_LEN"] = str(seg_len)
    else:
        os.environ.pop("PSDR_SEG_LEN", None)
    ctx = Context(N, True, 12 if log2n == 22 else 11, input_format="s16", max_batch=F)
    d = ctx.dev_alloc(raw.nbytes)
    ctx.h2d(d, raw)
    hb = ctx.half_frame_bytes()
    fn = _lib.load().psdr_debug_seg_fallbacks
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_void_p, C.c_uint]
    out, stats = {}, []
    for o in offsets:
        ctx.process_batch(d, F, offset_bytes=o * hb)
        ns, fb = C.c_uint(0), C.c_uint(0)
        assert fn(ctx.h, C.byref(ns), C.byref(fb), None, 0) == 0
        stats.append((ns.value, fb.value))
        for f in range(F):
            out[(o, f)] = hashlib.blake2b(ctx.read_quantized(f).tobytes(), digest_size=12).digest()
    ctx.dev_free(d)
    ctx.close()
    return out, stats


offs = [int(x) for x in rng.integers(0, NH - F, size=launches)]
got, stats = digests(0, offs)
uniq = sorted(set(offs))
ref, _ = digests(G, uniq)
# frame f of the launch at half-frame o is frame f + (o - o2) of a launch at o2: compare through the reference of its own offset
bad = [(o, f) for (o, f) in got if got[(o, f)] != ref[(o, f)]]
print({"log2n": log2n, "launches": launches, "frames_compared": len(got), "mismatches": len(bad), "first": bad[:5],
       "segments_and_fallbacks": stats[:8]})
sys.exit(1 if bad else 0)
