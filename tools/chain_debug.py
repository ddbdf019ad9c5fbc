#!/usr/bin/env python
"""debug: fused chain demodulation against the two-kernel path on the same samples (run on the GPU box)"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import quantize_raw, synth_stream  # noqa: E402


def run(chain, k, N=1 << 17, F=int(os.environ.get("DBG_F", "7")), n=int(os.environ.get("DBG_N", "360")), nb=2):
    os.environ["PSDR_DEMOD_CHAIN"] = "1" if chain else "0"
    if k:
        os.environ["PSDR_DEMOD_K"] = str(k)
    else:
        os.environ.pop("PSDR_DEMOD_K", None)
    from phantomsdr_amd import SpectrumEngine
    eng = SpectrumEngine(4369067, N, False, input_format="s16", max_batch=F, max_clients=8, audio_sps=12000 if n == 360 else 24000)
    assert eng.params["audio_fft_size"] == n, eng.params
    x = synth_stream((nb * F + 1) * (N // 2), False, seed=5, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    R = N
    cl = [eng.add_audio_client(1000 + 3000 * i, 1000 + 3000 * i + (0 if m in ("USB",) else 120), 1000 + 3000 * i + 240, m)
          for i, m in enumerate(["USB", "LSB", "AM", "FM", "FM", "AM"])]
    d = eng.ctx.dev_alloc(raw.nbytes)
    eng.ctx.h2d(d, raw)
    hb = eng.ctx.half_frame_bytes()
    out = []
    for b in range(nb):
        eng.ctx.process_batch(d, F, offset_bytes=b * F * hb)
        eng.ctx.demod_batch(b * F)
        eng.ctx.synchronize()
        out.append([c.read_audio(F) for c in cl])
    eng.close()
    return out


ref = run(False, 0)
for k in (0, 1, 2, 3):
    got = run(True, k)
    for b in range(len(ref)):
        for ci in range(len(ref[b])):
            for name, u, v in zip(("audio", "pwr", "nan"), ref[b][ci], got[b][ci]):
                u, v = np.asarray(u), np.asarray(v)
                bad = np.argwhere(u.view(np.uint32) != v.view(np.uint32)) if u.dtype == np.float32 else np.argwhere(u != v)
                if len(bad):
                    print(f"K={k} batch {b} client {ci} {name}: {len(bad)} of {u.size} differ; first {bad[:4].tolist()} ref {u[tuple(bad[0])]} got {v[tuple(bad[0])]}; frames {sorted(set(bad[:, 0].tolist()))[:8]} cols {sorted(set(bad[:, -1].tolist()))[:6]}..{max(bad[:, -1])}")
    print("K", k, "done")
