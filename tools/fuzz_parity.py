#!/usr/bin/env python
"""Randomised parity fuzz: random transform sizes, sample formats, batch splits, client slices
(all modes, edges, empty and widest slices, odd/fractional mids), mode switches, clients that sit out
whole batches (psdr_client_set_paused: the oracle's client gets no send_audio call, src/websocket.cpp:170-176)
and waterfall windows, every frame compared with the oracle the way tests/test_gpu_parity.py does.
    tools/fuzz_parity.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import check_fm, pwr_tolerance, quantize_raw, rel_err, rel_l2, synth_stream  # noqa: E402
from oracle import oracle as O  # noqa: E402
from phantomsdr_amd import AudioClient, Context, WaterfallClient  # noqa: E402

MODES = ["USB", "LSB", "AM", "FM"]


def levels_for(R, waterfall_size=1024):
    lv, cur = 0, R
    while cur >= waterfall_size:
        lv += 1
        cur //= 2
    return max(lv, 1)


def one_case(rng, case):
    # mostly small transforms; every fourth case one of the big plans (1024-point row pass: tile-major
    # spectrum lines, fused real-input pass 2 with chain segments)
    m = int(rng.integers(12, 18)) if rng.random() < 0.75 else int(rng.integers(19, 22))
    is_real = bool(rng.integers(0, 2))
    N = 1 << (m + (1 if is_real else 0))
    R = N // 2 if is_real else N
    fmt = str(rng.choice(["u8", "s8", "u16", "s16", "f32", "f64"]))
    n = int(rng.choice([8, 60, 124, 248, 360, 720, 1000]))
    n = min(n, R // 4 // 4 * 4)
    levels = levels_for(R)
    splits = [int(x) for x in rng.integers(1, 7, size=int(rng.integers(1, 4)))]
    F = max(splits)
    nframes = sum(splits)
    skip = int(rng.integers(1, 4))
    x = synth_stream((nframes + 1) * (N // 2), is_real, seed=int(rng.integers(1 << 30)), fft_size=N)
    raw = quantize_raw(x, fmt, is_real)
    conv = O.convert(raw, fmt)
    halves = (conv if is_real else conv.view(np.complex64)).reshape(nframes + 1, N // 2)
    ncl = int(rng.integers(1, 7)) if rng.random() < 0.9 else int(rng.integers(60, 72))  # (now and then more than one group of 64 slots)
    # half of the cases: the post chain on (DC blocker + AGC + int16, bit-exact against the oracle's chain fed with the GPU's
    # float audio), at audio rates that give it other delays / look-aheads (D = rate / 750 * 2: 32, 8, 20, 58 - the last
    # not a multiple of 4: the scalar gather / output kernels; L = rate / 5)
    post = rng.random() < 0.5
    rate = int(rng.choice([12000, 3000, 8000, 22050, 48000, 24000, 6000])) if post else 12000  # (D = 128, 64, 16: the LDS-ring kernel)
    clients = []
    for _ in range(ncl):
        mode = MODES[int(rng.integers(0, 4))]
        kind = int(rng.integers(0, 6))
        w = int(rng.integers(0, n // 2 + 1))
        if kind == 0:      # lower edge
            mid = int(rng.integers(0, 3))
        elif kind == 1:    # upper edge
            mid = R - 1 - int(rng.integers(0, 3))
        elif kind in (2, 3) and R >= 8192:
            # around the rows the big plans treat specially: multiples of 1024 / 2048 (row 0 of an output
            # column, the seam between columns) and the half-way rows (row M1/2, the mirror axis)
            step = 2048 if rng.random() < 0.5 else 1024
            mid = int(rng.integers(1, R // step - 1)) * step + (step // 2 if kind == 3 else 0) + int(rng.integers(-3, 4))
        else:
            mid = int(rng.integers(n, R - n))
        if mode == "USB":
            l, r = mid, mid + w
        elif mode == "LSB":
            l, r = mid - w, mid
        else:
            l, r = mid - w // 2, mid + w // 2
        l, r = max(0, l), min(R - 1, r)
        if l > r:
            l = r
        midf = float(mid) + float(rng.choice([0.0, 0.25, 0.5, 0.75]))
        clients.append((mode, l, midf, r))
    desc = f"case {case}: N=2^{int(np.log2(N))} real={int(is_real)} fmt={fmt} n={n} splits={splits} skip={skip} post={int(post)} rate={rate} clients={clients[:8]}"
    ctx = Context(N, is_real, levels, additional_size=n, audio_fft_size=n, audio_rate=rate, input_format=fmt,
                  max_batch=F, max_clients=ncl, max_waterfall_clients=3, skip_num=skip)
    try:
        if post:
            # (round 6: now and then round 5's form of the chain, and / or its PCM as int16 rows - psdr_read_pcm delivers int32 either
            # way; drawn from a generator of its own so that the cases themselves are the ones of earlier rounds)
            r6 = np.random.default_rng(600000 + case)
            if r6.random() < 0.3:
                ctx.set_option(ctx.OPT_POST_CHAIN_AGC, 0)
            if r6.random() < 0.3:
                ctx.set_option(ctx.OPT_POST_CHAIN_PCM16, 1)
            ctx.set_post_chain(True)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        gcl, ocl = [], []
        chains = [O.PostChain(rate) for _ in clients]
        for mode, l, mid, r in clients:
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            ok_g = g.on_window_message(l, mid, r)
            o = O.AudioClient(is_real, n, rate, R)
            o.set_audio_demodulation(mode)
            ok_o = o.on_window_message(l, mid, r)
            assert bool(ok_g) == bool(ok_o), desc
            if not ok_o:  # rejected window: both keep their defaults; give both a valid one
                g.set_audio_range(n, float(n), n + 4)
                o.set_audio_range(n, float(n), n + 4)
            gcl.append(g)
            ocl.append(o)
        wcl = []
        for _ in range(int(rng.integers(0, 3))):
            wv = WaterfallClient(ctx)
            lv = int(rng.integers(0, levels))
            span = R >> lv
            l = int(rng.integers(0, max(1, span - 8)))
            r = int(rng.integers(l + 1, min(span, l + 3000) + 1))
            wv.set_waterfall_range(lv, l, r)
            wcl.append(wv)
        fo = O.FFT(N, is_real, levels, 0, n)
        hb = ctx.half_frame_bytes()
        frame = 0
        pause_case = rng.random() < 0.34
        for bi, nf in enumerate(splits):
            if bi == 1 and rng.integers(0, 2):  # a mode switch between batches
                ci = int(rng.integers(0, ncl))
                mode = MODES[int(rng.integers(0, 4))]
                chains[ci].reset_agc()  # (every demodulation message resets the AGC, src/signal.cpp:316-328)
                gcl[ci].set_audio_demodulation(mode)
                ocl[ci].set_audio_demodulation(mode)
            # a third of the cases: every client sits out this batch with probability 1/4
            paused = [pause_case and rng.random() < 0.25 for _ in gcl]
            for g, pz in zip(gcl, paused):
                g.set_paused(pz)
            ctx.process_batch(d, nf, offset_bytes=frame * hb)
            ctx.demod_batch(frame)
            ctx.waterfall_batch(frame)
            got = [None if pz else g.read_audio(nf) for g, pz in zip(gcl, paused)]
            pcm = [None if (pz or not post) else g.read_pcm(nf) for g, pz in zip(gcl, paused)]
            sent = [f for f in range(nf) if (frame + f) % skip == 0]
            wgot = [w.read_waterfall() for w in wcl]
            for f in range(nf):
                fo.load(halves[frame], halves[frame + 1])
                fo.execute()
                spec = fo.output().copy()
                Xg = ctx.read_spectrum(f)
                nb = N // 2 if is_real else N
                assert rel_err(Xg[:nb], spec[:nb]) < 1e-4, desc + f" frame {frame}: spectrum"
                qg, qo = ctx.read_quantized(f), fo.quantized()
                dq = np.abs(qg.astype(np.int16) - qo.astype(np.int16))
                assert dq.max() <= 1 and (dq != 0).mean() <= 2e-3, desc + f" frame {frame}: pyramid {dq.max()} {(dq != 0).mean()}"
                for ci, o in enumerate(ocl):
                    if paused[ci]:
                        continue
                    a_o, p_o, _, dropped = o.send_audio(spec, frame, fft=fo)
                    a_g, p_g, nan_g = got[ci][0][f], got[ci][1][f], got[ci][2][f]
                    tag = desc + f" client {ci} frame {frame}"
                    assert not dropped and nan_g == 0, tag
                    if post:
                        want = chains[ci].process(a_g)
                        assert np.array_equal(pcm[ci][f], want), tag + f": pcm, {np.count_nonzero(pcm[ci][f] != want)} of {want.size} differ"
                    assert abs(p_g - p_o) <= pwr_tolerance(p_o, o.fwd_scale), tag + f": pwr {p_g} vs {p_o}"
                    scale = max(np.abs(a_o).max(), 1e-30)
                    if o.mode == O.FM:
                        check_fm(a_g, a_o, o.baseband()[: o.n // 2], o.bb_prev, tag, fwd_scale=max(o.fwd_scale, o.fwd_scale_prev))
                    else:
                        # relative to the larger of the audio's own peak and the slice's amplitude sqrt(w*pwr):
                        # a one-bin slice whose phase sits near +-90 degrees demodulates (c2r: only Re of
                        # bin 0 counts) to a value far below |X|, and 1e-4 of |X| is the accuracy asked for
                        floor = 1e-4 * float(np.sqrt(max(p_o, 0.0) * max(o.r - o.l, 1)))
                        assert np.abs(a_g - a_o).max() <= 3e-4 * scale + floor + 1e-12, tag + f" {np.abs(a_g - a_o).max() / scale:.2e}"
                if f in sent:
                    si = sent.index(f)
                    for w, (rows, label) in zip(wcl, wgot):
                        assert np.array_equal(rows[si], ctx.quantized_level(qg, w.level)[w.l:w.r]), desc + " waterfall"
                frame += 1
        ctx.dev_free(d)
    finally:
        ctx.close()
    return desc


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    for c in range(cases):
        d = one_case(rng, c)
        print("ok", d[:150])
    print(f"fuzz: {cases} cases passed (seed {seed})")


if __name__ == "__main__":
    main()
