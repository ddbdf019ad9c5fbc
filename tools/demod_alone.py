import os, sys, json, numpy as np
sys.path.insert(0, "/root/repo")
from phantomsdr_amd import SpectrumEngine
C = int(sys.argv[1]); F = 64; N = 1 << 20
eng = SpectrumEngine(35_000_000, N, False, input_format="s16", max_batch=F, max_clients=C, max_waterfall_clients=4)
hb = eng.ctx.half_frame_bytes()
rng = np.random.default_rng(0)
raw = rng.integers(-64, 64, size=(F + 1) * hb // 2, dtype=np.int16)
eng.upload_ring(raw)
R = eng.params["fft_result_size"]
for i in range(C):
    m = int(rng.uniform(0.05 * R, 0.95 * R))
    eng.add_audio_client(m, float(m), m + 89, "USB" if i % 2 == 0 else "LSB")
eng.step(0, F); eng.ctx.synchronize()
eng.ctx.set_profiling(True); eng.ctx.reset_kernel_stats()
for i in range(10):
    eng.ctx.demod_batch(i * F)
eng.ctx.synchronize()
st = eng.ctx.kernel_stats()
print(C, {k: round(ms / cnt * 1e3, 1) for k, (ms, cnt) in st.items()})
