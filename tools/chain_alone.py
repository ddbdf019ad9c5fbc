#!/usr/bin/env python
"""The post chain's kernels WITHOUT the FFT passes beside them: one batch is transformed, then psdr_demod_batch is repeated on
its spectrum (demodulation + chain only).  Under rocprofv3 --kernel-trace --stats this gives the kernels' own durations on
an otherwise idle chip (clock uncapped) - what bounds k_pc_agc / k_pc_ma2 themselves.
    tools/chain_alone.py [clients] [F] [repeats]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phantomsdr_amd import SpectrumEngine  # noqa: E402

ncl = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
N = 1 << 20
raw = np.random.default_rng(5).integers(-3000, 3000, size=(F + 1) * N, dtype=np.int16)
eng = SpectrumEngine(35_000_000, N, False, input_format="s16", max_batch=F, max_clients=ncl, audio_sps=12000)
if os.environ.get("PSDR_BENCH_AGC_FORM") is not None:
    eng.ctx.set_option(eng.ctx.OPT_POST_CHAIN_AGC, int(os.environ["PSDR_BENCH_AGC_FORM"]))
eng.ctx.set_post_chain(True)
R = eng.params["fft_result_size"]
r2 = np.random.default_rng(9)
for i in range(ncl):
    m = int(r2.uniform(0.05 * R, 0.95 * R))
    mode = ("USB", "LSB", "AM", "FM")[i % 4]
    l, r = (m, m + 89) if mode == "USB" else (m - 89, m) if mode == "LSB" else (m - 149, m + 149)
    eng.add_audio_client(l, float(m), r, mode)
d = eng.ctx.dev_alloc(raw.nbytes)
eng.ctx.h2d(d, raw)
eng.ctx.process_batch(d, F)
eng.ctx.synchronize()
for b in range(reps):
    eng.ctx.demod_batch(b * F)
    eng.ctx.synchronize()
print("done")
eng.close()
