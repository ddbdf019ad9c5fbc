#!/usr/bin/env python
"""Post-chain pipeline soak: nb batches of F frames of 2^20-point IQ enqueued back to back (nothing read, nothing drained in
between: every batch takes the buffer set of the batch three before it while that one's chain may still be running), then
the last batch's PCM.  Repeated; and once with a full synchronisation after every call.  All runs must agree bit for bit.
    tools/soak_post.py [clients] [F] [nb] [repeats]"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantomsdr_amd import SpectrumEngine  # noqa: E402

ncl = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 40
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
N = 1 << 20
rng = np.random.default_rng(5)
ring_batches = 4
raw = rng.integers(-3000, 3000, size=(ring_batches * F + 1) * N, dtype=np.int16)


def run(drained):
    eng = SpectrumEngine(35_000_000, N, False, input_format="s16", max_batch=F, max_clients=ncl, audio_sps=12000)
    try:
        eng.ctx.set_post_chain(True)
        R = eng.params["fft_result_size"]
        r2 = np.random.default_rng(9)
        cl = []
        for i in range(ncl):
            m = int(r2.uniform(0.05 * R, 0.95 * R))
            mode = ("USB", "LSB", "AM", "FM")[i % 4]
            l, r = (m, m + 89) if mode == "USB" else (m - 89, m) if mode == "LSB" else (m - 44, m + 44)
            cl.append(eng.add_audio_client(l, float(m), r, mode))
        d = eng.ctx.dev_alloc(raw.nbytes)
        eng.ctx.h2d(d, raw)
        hb = eng.ctx.half_frame_bytes()
        for b in range(nb):
            eng.ctx.process_batch(d, F, offset_bytes=(b % ring_batches) * F * hb)
            if drained:
                eng.ctx.synchronize()
            eng.ctx.demod_batch(b * F)
            if drained:
                eng.ctx.synchronize()
        crc = 0
        nz = 0
        for c in cl:
            pcm = c.read_pcm(F)
            crc = zlib.crc32(np.ascontiguousarray(pcm).tobytes(), crc)
            nz += int(np.count_nonzero(pcm))
        eng.ctx.dev_free(d)
        return crc, nz
    finally:
        eng.close()


ref = run(True)
assert ref[1] > 0, "the AGC never opened"
bad = 0
for i in range(reps):
    got = run(False)
    bad += got != ref
    print(f"run {i}: crc {got[0]:08x} nonzero {got[1]} {'ok' if got == ref else 'MISMATCH against the drained run ' + format(ref[0], '08x')}")
print(f"soak_post: {ncl} clients, {nb} batches of {F} frames, {reps} piped runs, {bad} mismatching")
sys.exit(1 if bad else 0)
