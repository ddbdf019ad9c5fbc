#!/usr/bin/env python
"""Plain step of the k-th context of a process (does the side stream ever get a hardware queue on the main stream's pipe?)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantomsdr_amd import SpectrumEngine  # noqa: E402
N, F = 1 << 20, 512
rng = np.random.default_rng(0)
keep = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    eng = SpectrumEngine(35_000_000, N, False, input_format="s16", max_batch=F, max_clients=16, max_waterfall_clients=4)
    hb = eng.ctx.half_frame_bytes()
    raw = rng.integers(-64, 64, size=(2 * F + 1) * hb // 2, dtype=np.int16)
    eng.upload_ring(raw)
    R = eng.params["fft_result_size"]
    for i in range(16):
        m = int(rng.uniform(0.05 * R, 0.95 * R))
        eng.add_audio_client(m, float(m), m + 89, "USB" if i % 2 == 0 else "LSB")
    for i in range(5):
        eng.step((i % 2) * F, F)
    eng.ctx.synchronize()
    eng.ctx.timer_start()
    for i in range(60):
        eng.step((i % 2) * F, F)
    ms = eng.ctx.timer_stop_ms() / 60
    print(f"context {k}: {ms:.3f} ms per step")
    if k % 3 == 2:
        keep.append(eng)  # (every third context stays alive: its streams keep their queues)
    else:
        eng.close()
