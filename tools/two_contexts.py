#!/usr/bin/env python
"""The post chain in the SECOND context of a process (bench.py's sub-workloads): which hardware queues do its streams get
once the first context's are gone?   tools/two_contexts.py [clients of the second]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantomsdr_amd import SpectrumEngine  # noqa: E402

N, F = 1 << 20, 512
ncl = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)


def run(clients, post, steps, close=True):
    eng = SpectrumEngine(35_000_000, N, False, input_format="s16", max_batch=F, max_clients=clients, max_waterfall_clients=4)
    if post:
        eng.ctx.set_post_chain(True)
    hb = eng.ctx.half_frame_bytes()
    raw = rng.integers(-64, 64, size=(2 * F + 1) * hb // 2, dtype=np.int16)
    eng.upload_ring(raw)
    R = eng.params["fft_result_size"]
    for i in range(clients):
        m = int(rng.uniform(0.05 * R, 0.95 * R))
        eng.add_audio_client(m, float(m), m + 89, "USB" if i % 2 == 0 else "LSB")
    for i in range(5):
        eng.step((i % 2) * F, F)
    eng.ctx.synchronize()
    eng.ctx.timer_start()
    for i in range(steps):
        eng.step((i % 2) * F, F)
    ms = eng.ctx.timer_stop_ms() / steps
    if close:
        eng.close()
    return ms


a = run(16, True, 20)
b = run(ncl, False, 60)
c = run(ncl, True, 120)
print(f"first context (16 clients, chain on) {a:.3f} ms; second: {ncl} clients plain {b:.3f} ms; third: chain on {c:.3f} ms (+{(c / b - 1) * 100:.1f} %)")
