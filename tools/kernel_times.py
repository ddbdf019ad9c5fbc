#!/usr/bin/env python
"""Per-kernel hipEvent timings of the cfg2-shaped hot path (used for ablations/tuning).
  PSDR_LIB=<variant .so> python tools/kernel_times.py [--fft 20] [--real] [--batch 16]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--fft", type=int, default=20)
ap.add_argument("--real", action="store_true")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--clients", type=int, default=16)
ap.add_argument("--ring-mib", type=int, default=512)
ap.add_argument("--tag", default="")
ap.add_argument("--post", action="store_true", help="enable the post-demodulation chain")
ap.add_argument("--audio-sps", type=int, default=12000, help="audio rate (the post chain's delays and look-ahead follow it)")
ap.add_argument("--mixed", action="store_true", help="clients alternate USB / LSB / AM / FM (default: USB / LSB)")
ap.add_argument("--mode", type=int, default=2, help="1: hipEvent brackets around every kernel; 2: device-clock stamps of the two passes")
args = ap.parse_args()

from phantomsdr_amd import SpectrumEngine  # noqa: E402

N, F = 1 << args.fft, args.batch
sps = 70_000_000 if args.real else 35_000_000
eng = SpectrumEngine(sps, N, args.real, input_format="s16", max_batch=F, max_clients=max(args.clients, 1),
                     max_waterfall_clients=4, audio_sps=args.audio_sps)
if args.post:
    if os.environ.get("PSDR_BENCH_AGC_FORM") is not None:  # (A/B of the chain's two AGC forms: psdr.h PSDR_OPT_POST_CHAIN_AGC)
        eng.ctx.set_option(eng.ctx.OPT_POST_CHAIN_AGC, int(os.environ["PSDR_BENCH_AGC_FORM"]))
    eng.ctx.set_post_chain(True)
hb = eng.ctx.half_frame_bytes()
nb = max(1, (args.ring_mib << 20) // (hb * F))
rng = np.random.default_rng(0)
raw = rng.integers(-64, 64, size=(nb * F + 1) * hb // 2, dtype=np.int16)
eng.upload_ring(raw)
R = eng.params["fft_result_size"]
for i in range(args.clients):
    m = int(rng.uniform(0.05 * R, 0.95 * R))
    md = ("USB", "LSB", "AM", "FM")[i % 4] if args.mixed else ("USB" if i % 2 == 0 else "LSB")
    w = 89 * args.audio_sps // 12000
    if md in ("AM", "FM"):
        eng.add_audio_client(m - w, float(m), m + w, md)
    else:
        eng.add_audio_client(m, float(m), m + w, md)
eng.add_waterfall_client()
for i in range(5):
    eng.step((i % nb) * F, F)
eng.ctx.synchronize()
eng.ctx.set_profiling(args.mode)
eng.ctx.reset_kernel_stats()
eng.ctx.timer_start()
for i in range(args.steps):
    eng.step((i % nb) * F, F)
total_ms = eng.ctx.timer_stop_ms()
st = eng.ctx.kernel_stats()
out = {"tag": args.tag or os.environ.get("PSDR_LIB", "default"), "N": N, "F": F,
       "us_per_frame_total": round(total_ms * 1e3 / (args.steps * F), 3)}
for k, (ms, cnt) in st.items():
    out[k] = round(ms / cnt * 1e3, 2)
if args.mode == 2:
    for k in ("fft_pass1", "fft_pass2"):
        out[k + "_median"] = round(float(np.median(eng.ctx.kernel_samples(k))), 2)
out["mode"] = args.mode
print(json.dumps(out))
eng.close()
