#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of F consecutive frames of a
device-resident raw sample ring: windowed 50 %-overlap forward FFT + int8 waterfall
pyramid, then the per-client slice -> inverse DFT -> demodulation for every audio client
and the byte gather for every waterfall client.  N=1 workload = BASELINE.json configs[1]:
35 MSPS-shape IQ cs16, 2^20-point FFT, 16 SSB audio clients + 4 waterfall clients.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` is whole-job ingest in MSamples/s (new complex
samples per second = frames/s * N/2) with the ring already resident in HBM.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md)

WORKLOADS = {
    # BASELINE.json configs[1]
    "cfg2": dict(sps=35_000_000, fft_size=1 << 20, is_real=False, fmt="s16", audio=16, waterfall=4,
                 modes=("USB", "LSB"), desc="35 MSPS IQ cs16, 2^20-pt C2C, 16 SSB audio + 4 waterfall clients"),
    # BASELINE.json configs[2]
    "cfg3": dict(sps=70_000_000, fft_size=1 << 21, is_real=True, fmt="s16", audio=64, waterfall=0,
                 modes=("USB", "LSB", "AM", "FM"), desc="70 MSPS real s16, 2^21-pt R2C, 64 mixed AM/FM/SSB clients"),
    # per-GPU share of BASELINE.json configs[3] (256 audio clients over 8 GPUs)
    "cfg4": dict(sps=35_000_000, fft_size=1 << 20, is_real=False, fmt="s16", audio=32, waterfall=0,
                 modes=("USB", "LSB", "AM", "FM"), desc="35 MSPS IQ cs16, 2^20-pt C2C, 32 audio clients per GPU"),
    # per-GPU share of BASELINE.json configs[4] (1024 clients + 64 zoomed waterfalls over 8 GPUs)
    "cfg5": dict(sps=70_000_000, fft_size=1 << 22, is_real=True, fmt="s16", audio=128, waterfall=8,
                 modes=("USB", "LSB", "AM", "FM"),
                 desc="70 MSPS real s16, 2^22-pt R2C, 128 audio clients + 8 zoomed waterfalls per GPU"),
}
SAMPLE_BYTES = {"u8": 1, "s8": 1, "u16": 2, "s16": 2, "f32": 4, "f64": 8}


def make_clients(wl, params, seed, count=None, first=0):
    """tuned ranges as in SURVEY 8d: centre uniform over the middle 90 % of [0,R); USB
    [m, m+3 kHz), LSB (m-3 kHz, m], AM/FM m +- 5 kHz; widths BW*N/sps bins."""
    rng = np.random.default_rng(seed)
    R = params["fft_result_size"]
    N, sps = wl["fft_size"], wl["sps"]
    b3 = int(3000 * N / sps)
    b5 = int(5000 * N / sps)
    out = []
    total = wl["audio"] if count is None else count
    for i in range(first + total):
        mode = wl["modes"][i % len(wl["modes"])]
        m = int(rng.uniform(0.05 * R, 0.95 * R))
        if mode == "USB":
            c = (mode, m, float(m), m + b3)
        elif mode == "LSB":
            c = (mode, m - b3, float(m), m)
        else:
            c = (mode, m - b5, float(m), m + b5)
        if i >= first:
            out.append(c)
    return out


def make_waterfalls(wl, params, seed):
    rng = np.random.default_rng(seed + 1000)
    R, levels = params["fft_result_size"], params["downsample_levels"]
    out = [(levels - 1, 0, R >> (levels - 1))]  # full span at the coarsest level
    for i in range(1, wl["waterfall"]):
        lv = int(rng.integers(0, levels - 1))
        width = 1024
        l = int(rng.integers(0, (R >> lv) - width))
        out.append((lv, l, l + width))
    return out[: wl["waterfall"]]


def algorithmic_bytes_per_frame(wl, params, clients, waterfalls):
    """B_frame = N*S_in + 8*R_spec + Q + C*(8*w + 4*n/2 + 4) + W*2*v"""
    N, is_real = wl["fft_size"], wl["is_real"]
    R = params["fft_result_size"]
    n = params["audio_fft_size"]
    s_in = SAMPLE_BYTES[wl["fmt"]] * (1 if is_real else 2)
    r_spec = N // 2 + 1 if is_real else N
    q = sum(R >> i for i in range(params["downsample_levels"]))
    b_in, b_spec, b_q = N * s_in, 8 * r_spec, q
    b_cl = sum(8 * (r - l) + 4 * (n // 2) + 4 for _, l, _, r in clients)
    b_wf = sum(2 * (r - l) for _, l, r in waterfalls) / params["skip_num"]
    return dict(input=b_in, spectrum=b_spec, pyramid=b_q, clients=b_cl, waterfall=b_wf,
                total=b_in + b_spec + b_q + b_cl + b_wf)


def gen_ring_torch(torch, device, nhalves, N, is_real, seed):
    """synthetic raw ring on the GPU: white Gaussian noise at sigma = 2^-9 FS plus 8 CW
    tones, quantised to s16 (interleaved I/Q for IQ)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    comps = 1 if is_real else 2
    half = (N // 2) * comps
    ring = torch.empty((nhalves, half), dtype=torch.int16, device=device)
    rng = np.random.default_rng(seed)
    tones = [(rng.uniform(-0.45, 0.45), rng.uniform(0.3, 1.0), rng.uniform(0, 6.28)) for _ in range(8)]
    amp = (2.0 if is_real else 1.0) / np.sqrt(N)
    for h in range(nhalves):
        t = torch.arange(h * (N // 2), (h + 1) * (N // 2), device=device, dtype=torch.float64)
        x = torch.randn((N // 2, comps), generator=g, device=device, dtype=torch.float32) * (2.0 ** -9)
        for f, a, ph in tones:
            arg = (2 * np.pi * f) * t + ph
            x[:, 0] += (amp * a * torch.cos(arg)).float()
            if not is_real:
                x[:, 1] += (amp * a * torch.sin(arg)).float()
        ring[h] = torch.clamp(torch.round(x * 32768.0), -32768, 32767).to(torch.int16).reshape(-1)
    return ring


def cpu_baseline(wl, params, clients, waterfalls, budget_s=15.0):
    """The oracle ("port") timed on this host's cores over a bounded sample of the same workload.
    Frames are independent for everything but the clients' overlap-add tails, so the CPU gets
    the same deal as the GPU's batches: W workers (threads; the oracle's C calls release the GIL),
    each a single-threaded pipeline - forward FFT + pyramid + every client's send_audio +
    the waterfall slices - on its own frames; `value` is the aggregate, `cores` = W (the best of
    a short probe over worker counts: the pipelines are memory-bound long before 256 threads)."""
    import threading
    from oracle import oracle as O
    N, is_real = wl["fft_size"], wl["is_real"]
    n, levels = params["audio_fft_size"], params["downsample_levels"]
    R = params["fft_result_size"]
    rng = np.random.default_rng(1)
    nh = 4
    if is_real:
        halves = (rng.standard_normal((nh, N // 2)) * 2.0 ** -9).astype(np.float32)
    else:
        halves = ((rng.standard_normal((nh, N // 2)) + 1j * rng.standard_normal((nh, N // 2))) * 2.0 ** -9).astype(np.complex64)
    O.set_threads(1)
    ncpu = os.cpu_count() or 1

    class Worker:
        def __init__(self):
            self.fo = O.FFT(N, is_real, levels, 0, n)
            self.ocl = []
            for mode, l, m, r in clients:
                c = O.AudioClient(is_real, n, 12000, R)
                c.set_audio_demodulation(mode)
                c.set_audio_range(l, m, r)
                self.ocl.append(c)
            self.frames = 0

        def run(self, stop_at):
            fo = self.fo
            while time.perf_counter() < stop_at:
                f = self.frames
                fo.load(halves[f % (nh - 1)], halves[f % (nh - 1) + 1])
                fo.execute()
                spec = fo.output()
                for c in self.ocl:
                    c.send_audio(spec, f, fft=fo)
                if f % params["skip_num"] == 0:
                    q = fo.quantized()
                    for lv, l, r in waterfalls:
                        off = sum(R >> t for t in range(lv))
                        _ = q[off + l: off + r].tobytes()
                self.frames += 1

    def measure(workers, seconds):
        for w in workers:
            w.frames = 0
        t0 = time.perf_counter()
        ths = [threading.Thread(target=w.run, args=(t0 + seconds,)) for w in workers]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        return sum(w.frames for w in workers), dt

    cand = sorted({c for c in (8, 32, 64, 128, ncpu) if c <= ncpu} | {min(ncpu, 8)})
    pool = [Worker() for _ in range(max(cand))]
    best_rate, cores = 0.0, cand[0]
    probe_s = min(2.0, budget_s / (2 * len(cand)))
    for c in cand:
        fr, dt = measure(pool[:c], probe_s)
        if fr / dt > best_rate:
            best_rate, cores = fr / dt, c
    frames, dt = measure(pool[:cores], budget_s / 2)
    msps = frames * (N // 2) / dt / 1e6
    return {"value": round(msps, 3), "unit": "MSamples/s", "cores": cores, "kind": "port",
            "sample": f"{frames} frames of the same workload in {dt:.1f} s: {cores} single-threaded pipelines "
                      f"(oracle/psdr_oracle.c) side by side on {ncpu} host threads, best worker count of a probe"}


def emit(out):
    """the ONE JSON line, as the last thing on stdout: RCCL prints a banner through C stdio whose
    buffer would otherwise be flushed after Python's at exit"""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if out is not None:
        print(json.dumps(out), flush=True)


def kernel_roofline(ctx, step, first_step, nsteps, wl, wl_name, params, clients, waterfalls, frames_per_launch):
    """per-kernel durations from a profiled replay (hipEvents on the library's own streams) and the
    roofline block of the dominant kernel (DESIGN.md "Roofline accounting")"""
    ctx.set_profiling(True)
    ctx.reset_kernel_stats()
    for i in range(nsteps):
        step(first_step + i)
    ctx.synchronize()
    stats = ctx.kernel_stats()
    ctx.set_profiling(False)
    ab = algorithmic_bytes_per_frame(wl, params, clients, waterfalls)
    Fl = frames_per_launch
    # algorithmic bytes by kernel: pass 1 reads the raw input once; pass 2 (+fused epilogue)
    # writes the spectrum and the pyramid; the demod kernels read slices and write audio;
    # intermediates count zero.
    per_kernel_bytes = {
        "fft_pass1": ab["input"] * Fl,
        "fft_pass2": (ab["spectrum"] + (ab["pyramid"] if not wl["is_real"] else 0)) * Fl,
        "untangle_real": (ab["spectrum"] + ab["pyramid"]) * Fl if wl["is_real"] else 0,
        "demod_idft": ab["clients"] * Fl,
    }
    kernels = {name: {"avg_us": round(ms / cnt * 1e3, 3), "launches": int(cnt)} for name, (ms, cnt) in stats.items()}
    dom = max(stats, key=lambda k: stats[k][0]) if stats else None
    roofline = None
    if dom:
        avg_s = stats[dom][0] / stats[dom][1] / 1e3
        achieved = per_kernel_bytes.get(dom, 0) / avg_s
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile)).get(wl_name, {})
                traffic = tj.get(dom)
                if traffic is not None:  # measured at tj["_frames_per_launch"] frames per launch
                    traffic = int(traffic * Fl / tj.get("_frames_per_launch", Fl))
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(per_kernel_bytes.get(dom, 0)),
                    "avg_launch_us": round(avg_s * 1e6, 2)}
    return roofline, kernels, ab


def run_sharded_bench(args, torch, rank, world, local_rank):
    """N > 1.  Default (--shard time): the STREAM is sharded - batch g goes to rank g mod G
    with a two-frame warm-up instead of any exchange (phantomsdr_amd/distributed.py); every
    rank serves all the clients of its frames; weak scaling (per-GPU work fixed), no data-path
    collective.  --shard clients: BASELINE.json configs[3] shape - audio clients sharded over
    the ranks, rank 0 FFTs and broadcasts each spectrum batch over RCCL/xGMI."""
    import torch.distributed as dist
    from phantomsdr_amd import SpectrumEngine
    from phantomsdr_amd.distributed import (HipBackend, HipTimeBackend, ShardedRunner, TimeShardedRunner,
                                            assign_clients)

    device = torch.device("cuda", local_rank)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    time_mode = args.shard == "time"
    wl = WORKLOADS[args.workload or ("cfg2" if time_mode else "cfg4")]
    N = wl["fft_size"]
    warm = TimeShardedRunner.WARMUP if time_mode else 0
    # time mode: a launch carries --batch frames INCLUDING the two warm-up frames, so its tile
    # count stays a multiple of the work-group count (258 frames would leave half the
    # work-groups one tile short: +20 us of tail per step); F = the NEW frames per step
    F = args.batch - warm
    per_gpu = wl["audio"]
    eng = SpectrumEngine(wl["sps"], N, wl["is_real"], input_format=wl["fmt"], max_batch=F + warm,
                         max_clients=max(per_gpu, 1), max_waterfall_clients=max(wl["waterfall"], 1),
                         device=local_rank)
    params = eng.params
    hb = eng.ctx.half_frame_bytes()
    nbatches = max(2, (args.ring_mib * (1 << 20)) // (hb * F))
    ring = None
    ring_ptr = 0
    if time_mode or rank == 0:
        ring = gen_ring_torch(torch, device, nbatches * F + 1, N, wl["is_real"], seed=0x5D5D0004)
        ring_ptr = ring.data_ptr()
    torch.cuda.synchronize()

    if time_mode:
        clients = make_clients(wl, params, seed=0x5D5D0002)
        waterfalls = make_waterfalls(wl, params, seed=0x5D5D0002)
        for mode, l, m, r in clients:
            eng.add_audio_client(l, m, r, mode)
        for lv, l, r in waterfalls:
            eng.add_waterfall_client(lv, l, r)
        backend = HipTimeBackend(eng.ctx, ring_ptr, nbatches * F + 1, F + warm)
        runner = TimeShardedRunner(backend, rank, world, F)

        def step(i):
            first, skip = runner.step(i)
            if waterfalls:
                eng.ctx.waterfall_batch(first - skip)
        nclients_total, par = len(clients), (
            f"stream sharded over {world} GPUs: batch g -> rank g mod G, 2-frame warm-up, no data-path collective")
        bytes_bcast = None
    else:
        all_clients = make_clients(wl, params, seed=0x5D5D0004, count=per_gpu * world)
        mine = assign_clients(len(all_clients), world)[rank]
        clients = [all_clients[c] for c in mine]
        waterfalls = []
        for mode, l, m, r in clients:
            eng.add_audio_client(l, m, r, mode)
        torch.cuda.synchronize()
        backend = HipBackend(torch, eng.ctx, device, ring_ptr, nbatches, F)
        runner = ShardedRunner(backend, dist, rank, world, F)
        step = runner.step
        nclients_total, par = per_gpu * world, (
            f"clients sharded over {world} GPUs (client i -> rank i mod G); rank 0 FFT + RCCL broadcast "
            "of the spectrum batch")

    for i in range(args.warmup):
        step(i)
    eng.ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    if not time_mode:
        runner.bytes_broadcast = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    eng.ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # time mode: every rank ingests its own F new frames per step; client mode: one stream
    frames = args.steps * F * (world if time_mode else 1)
    msps = frames * (N // 2) / dt / 1e6
    # roofline of the dominant kernel on rank 0's GPU (all ranks replay: client mode broadcasts)
    wl_name_s = args.workload or ("cfg2" if time_mode else "cfg4")
    roofline, kernels, _ = kernel_roofline(eng.ctx, step, args.warmup + args.steps, min(args.steps, 20), wl,
                                           wl_name_s, params, clients, waterfalls, F + warm)
    eng.ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        ab = algorithmic_bytes_per_frame(wl, params, clients, waterfalls)
        out = {
            "metric": "ingest MSamples/s + concurrent audio clients at 2^20-pt FFT",
            "value": round(msps, 2), "unit": "MSamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (args.workload or ("cfg2" if time_mode else "cfg4")) + ": " + wl["desc"],
                       "frames_per_step": F, "fft_size": N, "audio_clients": nclients_total,
                       "waterfall_clients": len(waterfalls), "parallelism": par,
                       "realtime_factor": round(msps * 1e6 / wl["sps"], 1)},
            "roofline": roofline,
            "path": {"algorithmic_bytes_per_frame": int(ab["total"]),
                     "frames_per_s": round(frames / dt, 1), "kernels": kernels,
                     "frac_of_hbm_peak_per_gpu": round(ab["total"] * frames / dt / HBM_PEAK / (world if time_mode else 1), 4)},
            "xgmi": None if time_mode else {
                "broadcast_bytes_per_frame": 8 * N,
                "GB_per_s_per_link": round(runner.bytes_broadcast / dt / 1e9, 2) if world > 1 else None,
                "link_peak_GB_per_s": 153.0},
            "cpu_baseline": None,
        }
    else:
        out = None
    eng.close()
    if dist.is_initialized():
        emit(None)  # every rank flushes what RCCL wrote through C stdio ...
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        time.sleep(0.2)  # ... and rank 0's line comes last
    emit(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per step (F); default 256: a step of the two persistent passes has ~90 us of "
                         "fixed cost (ramp, prologue, tail, launch gaps) whatever F is (DESIGN.md, batch-size table)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-post-chain", action="store_true",
                    help="skip the separate post-demodulation-chain measurement (used for the rocprofv3 kernel stats: "
                         "its passes overlap the long chain kernels and would skew the per-kernel averages)")
    ap.add_argument("--ring-mib", type=int, default=512)
    ap.add_argument("--shard", default="time", choices=["time", "clients"],
                    help="N > 1: shard the stream (default) or the clients (spectrum broadcast)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path even with one rank (testing)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if args.batch <= 0:
        args.batch = 256

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    if world > 1 or args.force_sharded:
        return run_sharded_bench(args, torch, rank, world, local_rank)

    from phantomsdr_amd import SpectrumEngine
    wl_name = args.workload or "cfg2"
    wl = WORKLOADS[wl_name]
    F = args.batch
    eng = SpectrumEngine(wl["sps"], wl["fft_size"], wl["is_real"], input_format=wl["fmt"],
                         max_batch=F, max_clients=max(wl["audio"], 1),
                         max_waterfall_clients=max(wl["waterfall"], 1), device=local_rank)
    params = eng.params
    N = wl["fft_size"]
    hb = eng.ctx.half_frame_bytes()
    # ring > 256 MiB Infinity Cache; a whole number of batches (+1 trailing half)
    nbatches = max(1, (args.ring_mib * (1 << 20)) // (hb * F))
    nhalves = nbatches * F + 1
    ring = gen_ring_torch(torch, device, nhalves, N, wl["is_real"], seed=0x5D5D0002)
    torch.cuda.synchronize()
    eng.ring = None
    ring_ptr = ring.data_ptr()

    clients = make_clients(wl, params, seed=0x5D5D0002)
    waterfalls = make_waterfalls(wl, params, seed=0x5D5D0002)
    for mode, l, m, r in clients:
        eng.add_audio_client(l, m, r, mode)
    for lv, l, r in waterfalls:
        eng.add_waterfall_client(lv, l, r)

    def step(i):
        b = i % nbatches
        eng.ctx.process_batch(ring_ptr, F, offset_bytes=b * F * hb)
        if clients:
            eng.ctx.demod_batch(eng.frame_num)
        if waterfalls:
            eng.ctx.waterfall_batch(eng.frame_num)
        eng.frame_num += F

    for i in range(args.warmup):
        step(i)
    eng.ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    eng.ctx.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    frames = args.steps * F
    msps = frames * (N // 2) / dt / 1e6
    ms_per_step = dt / args.steps * 1e3

    # per-kernel durations: second, profiled replay of the same steps (the events do not perturb `value`)
    roofline, kernels, ab = kernel_roofline(eng.ctx, step, args.warmup, min(args.steps, 50), wl, wl_name, params,
                                            clients, waterfalls, F)
    path_frac = ab["total"] * (frames / dt) / HBM_PEAK

    # SURVEY 8f-2 (widened row): the optional post-demodulation chain (DC blocker + AGC + int16),
    # measured separately - it is NOT part of `value` (the metric's clients end at float audio)
    post = None
    if clients and not args.no_post_chain:
        try:
            eng.ctx.set_post_chain(True)
            psteps = 6
            for i in range(2):
                step(i)
            eng.ctx.synchronize()
            t0 = time.perf_counter()
            for i in range(psteps):
                step(2 + i)
            eng.ctx.synchronize()
            pdt = (time.perf_counter() - t0) / psteps
            eng.ctx.set_post_chain(False)
            h = params["audio_fft_size"] // 2
            post = {"ms_per_step": round(pdt * 1e3, 3), "MSamples_per_s_ingest": round(F * (N // 2) / pdt / 1e6, 1),
                    "audio_samples_per_s": round(len(clients) * F * h / pdt, 1),
                    "realtime_factor": round(F * (N // 2) / pdt / wl["sps"], 1),
                    "note": "whole step with psdr_set_post_chain(1): f32 recurrences, sequential per client"}
        except Exception as e:
            post = {"error": repr(e)}

    cpu = None
    if not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(wl, params, clients, waterfalls)
        except Exception as e:  # the oracle is optional for the measured value
            cpu = {"error": repr(e)}

    out = {
        "metric": "ingest MSamples/s + concurrent audio clients at 2^20-pt FFT",
        "value": round(msps, 2), "unit": "MSamples/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl_name + ": " + wl["desc"], "frames_per_step": F,
                   "fft_size": N, "audio_clients": len(clients), "waterfall_clients": len(waterfalls),
                   "audio_fft_size": params["audio_fft_size"], "ring_MiB": round(nhalves * hb / 2 ** 20, 1),
                   "realtime_factor": round(msps * 1e6 / wl["sps"], 1)},
        "roofline": roofline,
        "path": {"algorithmic_bytes_per_frame": int(ab["total"]), "frames_per_s": round(frames / dt, 1),
                 "frac_of_hbm_peak": round(path_frac, 4), "kernels": kernels},
        "post_chain": post,
        "cpu_baseline": cpu,
    }
    eng.close()
    emit(out)


if __name__ == "__main__":
    main()
