import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np
from phantomsdr_amd import SpectrumEngine
N, F = 1 << 20, 256
eng = SpectrumEngine(35_000_000, N, False, input_format="s16", max_batch=F, max_clients=16, max_waterfall_clients=4)
hb = eng.ctx.half_frame_bytes()
raw = np.random.default_rng(0).integers(-64, 64, size=(F * 2 + 1) * hb // 2, dtype=np.int16)
eng.upload_ring(raw)
R = eng.params["fft_result_size"]
rng = np.random.default_rng(1)
for i in range(16):
    m = int(rng.uniform(0.1 * R, 0.9 * R)); eng.add_audio_client(m, float(m), m + 90, "USB")
for post in (False, True):
    eng.ctx.set_post_chain(post)
    for i in range(6): eng.step((i % 2) * F, F)
    eng.ctx.synchronize()
    t0 = time.perf_counter(); hs = []
    for i in range(40):
        a = time.perf_counter(); eng.step((i % 2) * F, F); hs.append(time.perf_counter() - a)
    t1 = time.perf_counter(); eng.ctx.synchronize(); t2 = time.perf_counter()
    print("post", post, "host enqueue per step us: median %.0f max %.0f; enqueue total %.1f ms, total %.1f ms -> %.3f ms/step" % (np.median(hs)*1e6, max(hs)*1e6, (t1-t0)*1e3, (t2-t0)*1e3, (t2-t0)*1e3/40))
eng.close()
