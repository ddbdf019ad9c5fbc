#!/usr/bin/env python
"""Runs ON THE GPU BOX.  The consumer kernels of a batch (real seam, column tail, pyramid tail, demodulation chain,
waterfall gather) timed with hipEvent brackets (psdr_set_profiling mode 1) in two schedules:
  alone   every step is drained before the next one starts - the consumers have the chip to themselves
  beside  steps back to back - the consumers of batch b run beside the passes of batch b+1 (what the bench times)
with the bench's own client sets (bench.make_clients).  One JSON line per workload.
  python tools/consumers_alone.py [cfg3 cfg5 clients256 ...] [--clients N] [--batch 256] [--steps 6]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from phantomsdr_amd import SpectrumEngine  # noqa: E402
from phantomsdr_amd.core import derived_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workloads", nargs="*", default=["cfg3", "cfg5", "clients256"])
ap.add_argument("--clients", type=int, default=0)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--tag", default="")
args = ap.parse_args()

for name in args.workloads:
    wl = dict(bench.WORKLOADS[name])
    if args.clients:
        wl["audio"] = args.clients
    N, F = wl["fft_size"], args.batch
    p = derived_params(wl["sps"], N, wl["is_real"])
    eng = SpectrumEngine(wl["sps"], N, wl["is_real"], input_format="s16", max_batch=F, max_clients=max(wl["audio"], 1),
                         max_waterfall_clients=max(wl["waterfall"], 1))
    hb = eng.ctx.half_frame_bytes()
    rng = np.random.default_rng(0)
    nb = 2
    raw = rng.integers(-64, 64, size=(nb * F + 1) * hb // 2, dtype=np.int16)
    eng.upload_ring(raw)
    for mode, l, m, r in bench.make_clients(wl, p, seed=1):
        eng.add_audio_client(l, m, r, mode)
    for lv, l, r in bench.make_waterfalls(wl, p, seed=1):
        eng.add_waterfall_client(lv, l, r)
    out = {"workload": name, "clients": wl["audio"], "n": p["audio_fft_size"], "F": F, "tag": args.tag or os.environ.get("PSDR_LIB", "default")}
    for sched in ("alone", "beside"):
        for i in range(3):
            eng.step((i % nb) * F, F)
        eng.ctx.synchronize()
        eng.ctx.set_profiling(1)
        eng.ctx.reset_kernel_stats()
        eng.ctx.timer_start()
        for i in range(args.steps):
            eng.step((i % nb) * F, F)
            if sched == "alone":
                eng.ctx.synchronize()
        ms = eng.ctx.timer_stop_ms()
        st = eng.ctx.kernel_stats()
        # microseconds per STEP (pyramid_tail = column tail + the levels above it: two launches)
        out[sched] = {k: round(t * 1e3 / args.steps, 1) for k, (t, c) in st.items()}
        out[sched]["step_us"] = round(ms * 1e3 / args.steps, 1)
        eng.ctx.set_profiling(0)
    print(json.dumps(out), flush=True)
    eng.close()
