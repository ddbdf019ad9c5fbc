"""What the reference does NOT do to a client's state, reproduced on the GPU and checked against the oracle:

* a client whose socket is backed up gets no send_audio call (src/websocket.cpp:170-176): overlap-add tails, FM's last
  sample, DC blocker and AGC stand still for the frame (src/signal.cpp:200-203, 273-284) - psdr_client_set_paused;
* a USB / LSB frame dropped by the NaN guard throws BEFORE audio_real_prev is replaced (src/signal.cpp:266-275): the
  next frame adds the tail of the last frame that survived.  AM / FM state moves before the guard (:200-203)."""
import ctypes as C

import numpy as np
import pytest

from helpers import check_fm, pwr_tolerance, quantize_raw, rel_l2, synth_stream
from oracle import oracle as O
from test_gpu_parity import levels_for

pytestmark = pytest.mark.gpu


def _compare(a_g, a_o, o, tag):
    if o.mode == O.FM:
        check_fm(a_g, a_o, o.baseband()[: o.n // 2], o.bb_prev, tag, fwd_scale=max(o.fwd_scale, o.fwd_scale_prev))
    else:
        assert rel_l2(a_g, a_o) < 1e-4, f"{tag}: rel L2 {rel_l2(a_g, a_o):.2e}"


@pytest.mark.parametrize("n,F,is_real", [(360, 6, 0), (248, 5, 0), (720, 9, 1), (360, 1, 0)])
def test_paused_clients_keep_their_state_like_the_reference(n, F, is_real):
    """Batches of F frames; clients sit out whole batches (psdr_client_set_paused).  The oracle's clients get no
    send_audio call for those frames.  Float audio against the oracle, PCM (post chain on) bit-exact against the
    oracle's chain fed with the GPU's float audio of the frames that were NOT skipped; a paused client's results read
    as PSDR_ERR_NO_DATA.  n = 360 / 720: the chain kernel, 248: the generic kernels."""
    from phantomsdr_amd import AudioClient, Context, PsdrError
    N = 1 << 15
    R = N // 2 if is_real else N
    nb = 10 if F > 1 else 40
    levels = levels_for(R)
    x = synth_stream((nb * F + 1) * (N // 2), bool(is_real), seed=91, fft_size=N)
    raw = quantize_raw(x, "s16", bool(is_real))
    conv = O.convert(raw, "s16")
    halves = (conv if is_real else conv.view(np.complex64)).reshape(nb * F + 1, N // 2)
    am = int(0.11 * N) if is_real else int((0.11 * N - (N // 2 + 1)) % N)
    fm = int(0.31 * N) if is_real else int((-0.21 * N - (N // 2 + 1)) % N)
    specs = [("USB", am, float(am), am + 80), ("LSB", am - 80, am + 0.5, am), ("AM", am - 100, float(am), am + 100),
             ("FM", fm - 100, fm + 0.5, fm + 100), ("AM", fm - 90, float(fm + 1), fm + 90)]
    # batch -> paused clients (client 4 is paused before its first batch ever runs; client 3 twice in a row and again)
    sched = {0: {4}, 1: {4, 1}, 2: {1, 3}, 3: {3}, 5: {0, 2}, 6: {3}, 8: {0, 1, 2, 3, 4}}
    if F == 1:
        sched = {b: set(c for c in range(5) if (b * 7 + c * 3) % 11 < 3) for b in range(nb)}
    ctx = Context(N, bool(is_real), levels, additional_size=n, audio_fft_size=n, audio_rate=12000, input_format="s16",
                  max_batch=F, max_clients=len(specs))
    try:
        ctx.set_post_chain(True)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        gcl, ocl, chains = [], [], []
        for mode, l, mid, r in specs:
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            g.set_audio_range(l, mid, r)
            gcl.append(g)
            o = O.AudioClient(bool(is_real), n, 12000, R)
            o.set_audio_demodulation(mode)
            o.set_audio_range(l, mid, r)
            ocl.append(o)
            chains.append(O.PostChain(12000))
        fo = O.FFT(N, bool(is_real), levels, 0, n)
        hb = ctx.half_frame_bytes()
        live = 0
        for b in range(nb):
            paused = sched.get(b, set())
            if b == 4:  # a mode change while nobody is paused, and one (client 1, batch 7) right after a pause
                gcl[2].set_audio_demodulation("FM"), ocl[2].set_audio_demodulation("FM"), chains[2].reset_agc()
            if b == 7:
                gcl[1].set_audio_demodulation("AM"), ocl[1].set_audio_demodulation("AM"), chains[1].reset_agc()
            for ci, g in enumerate(gcl):
                g.set_paused(ci in paused)
            ctx.process_batch(d, F, offset_bytes=b * F * hb)
            ctx.demod_batch(b * F)
            got = {}
            for ci, g in enumerate(gcl):
                if ci in paused:
                    with pytest.raises(PsdrError) as e:
                        g.read_audio(F)
                    assert e.value.code == -7
                    with pytest.raises(PsdrError) as e:
                        g.read_pcm(F)
                    assert e.value.code == -7
                else:
                    got[ci] = g.read_audio(F) + (g.read_pcm(F),)
            for f in range(F):
                frame = b * F + f
                fo.load(halves[frame], halves[frame + 1])
                fo.execute()
                spec = fo.output().copy()
                for ci, o in enumerate(ocl):
                    if ci in paused:
                        continue  # src/websocket.cpp:174-176: no send_audio call
                    a_o, p_o, _, dropped = o.send_audio(spec, frame, fft=fo)
                    a_g, p_g, nan_g, pcm_g = (got[ci][k][f] for k in range(4))
                    tag = f"client {ci} {specs[ci][0]} batch {b} frame {f} (paused this batch: {sorted(paused)})"
                    assert not dropped and nan_g == 0, tag
                    assert abs(p_g - p_o) <= pwr_tolerance(p_o, o.fwd_scale), tag
                    _compare(a_g, a_o, o, tag)
                    want = chains[ci].process(a_g)
                    assert np.array_equal(pcm_g, want), f"{tag}: PCM {np.count_nonzero(pcm_g != want)} of {want.size} differ"
                    live += int(np.count_nonzero(want) > 0)
        assert live > 20, "the AGC never opened"
        ctx.dev_free(d)
    finally:
        ctx.close()


@pytest.mark.parametrize("n,F,chain", [(360, 12, 1), (360, 12, 0), (720, 7, 1), (248, 6, 1)])
def test_ssb_tail_survives_dropped_frames_like_the_reference(n, F, chain, monkeypatch):
    """NaN samples in the INPUT (f32 format) poison the two frames that contain the half-frame - for every client.
    USB / LSB: the reference keeps audio_real_prev from the last frame that survived (the guard throws before
    src/signal.cpp:273-275), so the frame after the dropped ones adds THAT tail; AM / FM: the baseband buffers were
    already replaced (:200-203), the frame after a dropped one is poisoned through its overlap half and dropped too.
    Poisoned halves sit inside a chain, across a chain boundary (K = 4: frames 3|4, 7|8), at the batch start and at the
    batch end (the carried state).  Oracle = the reference's flow, frame by frame."""
    from phantomsdr_amd import AudioClient, Context
    monkeypatch.setenv("PSDR_DEMOD_CHAIN", str(chain))
    monkeypatch.setenv("PSDR_DEMOD_K", "4")
    N, nb = 1 << 14, 4
    levels = levels_for(N)
    nframes = nb * F
    x = synth_stream((nframes + 1) * (N // 2), False, seed=5, fft_size=N).astype(np.complex64)
    halves = x.reshape(nframes + 1, N // 2).copy()
    # half h is part of frames h-1 and h
    bad_halves = [2, 4, 5, 8, F, 2 * F - 1 + 1, 3 * F + 3, 3 * F + 4, 3 * F + 5]
    for h in bad_halves:
        halves[h, 17] = np.nan
    bad_frames = sorted({h - 1 for h in bad_halves if h >= 1} | {h for h in bad_halves if h < nframes})
    raw = halves.reshape(-1).view(np.float32).copy()
    am = int((0.11 * N - (N // 2 + 1)) % N)
    specs = [("USB", am, float(am), am + 80), ("LSB", am - 80, am + 0.5, am), ("AM", am - 100, float(am), am + 100),
             ("FM", am - 100, am + 0.5, am + 100)]
    ctx = Context(N, False, levels, additional_size=n, audio_fft_size=n, audio_rate=12000, input_format="f32",
                  max_batch=F, max_clients=len(specs))
    try:
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        gcl, ocl = [], []
        for mode, l, mid, r in specs:
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            g.set_audio_range(l, mid, r)
            gcl.append(g)
            o = O.AudioClient(False, n, 12000, N)
            o.set_audio_demodulation(mode)
            o.set_audio_range(l, mid, r)
            ocl.append(o)
        fo = O.FFT(N, False, levels, 0, n)
        hb = ctx.half_frame_bytes()
        ndropped = [0] * len(specs)
        nrecovered = 0
        for b in range(nb):
            ctx.process_batch(d, F, offset_bytes=b * F * hb)
            ctx.demod_batch(b * F)
            got = [g.read_audio(F) for g in gcl]
            for f in range(F):
                frame = b * F + f
                fo.load(halves[frame], halves[frame + 1])
                fo.execute()
                spec = fo.output().copy()
                for ci, o in enumerate(ocl):
                    a_o, p_o, _, dropped = o.send_audio(spec, frame, fft=fo)
                    a_g, nan_g = got[ci][0][f], got[ci][2][f]
                    tag = f"client {ci} {specs[ci][0]} frame {frame} (poisoned frames {bad_frames})"
                    assert bool(nan_g) == bool(dropped), tag
                    if dropped:
                        ndropped[ci] += 1
                        continue
                    _compare(a_g, a_o, o, tag)
                    nrecovered += int(ci < 2 and frame - 1 in bad_frames)
        # every poisoned frame is dropped for everybody; the complex modes lose the frame after a poisoned run as well
        assert ndropped[0] == ndropped[1] == len(bad_frames)
        assert ndropped[2] > ndropped[0] and ndropped[3] > ndropped[0]
        assert nrecovered >= 4  # SSB frames right after a dropped one matched the oracle: the surviving tail was added
        ctx.dev_free(d)
    finally:
        ctx.close()


def _eq_nan(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("n,chain", [(360, 1), (360, 0), (720, 1), (248, 1)])
def test_inf_and_nan_bins_batched_guard_equals_the_frame_by_frame_rule(n, chain, monkeypatch):
    """The NaN guard of USB / LSB is a recurrence over a client's frames (src/signal.cpp:266-275): frame f is dropped iff
    y_f[0..n/2) + audio_real_prev has a NaN, and audio_real_prev moves on only if it was not.  With ONE frame per batch the
    kernels evaluate exactly that (the carried tail IS audio_real_prev).  Spectra with +-Inf and NaN bins inside the
    clients' slices - transforms that are NaN in some outputs and +-Inf in others, tails that go bad and stay bad, Inf - Inf
    in the overlap-add - through psdr_demod_batch_from: batches of 5, 8 and 37 frames (chains of 4 with a warm-up frame,
    or the two-kernel path) must give bit for bit what the frame-by-frame run gives: audio, pwr, flags, every frame, and
    with them the state every later frame starts from.  (Round 5 documented an Inf sample as a deviation; the slots
    that see a non-finite value are now walked a second time in frame order: demod.h, DemodArgs::ssb_mark.)"""
    from phantomsdr_amd import AudioClient, Context
    monkeypatch.setenv("PSDR_DEMOD_CHAIN", str(chain))
    monkeypatch.setenv("PSDR_DEMOD_K", "4")
    N, nframes = 1 << 14, 74
    levels = levels_for(N)
    rng = np.random.default_rng(77)
    spec = ((rng.standard_normal((nframes, N)) + 1j * rng.standard_normal((nframes, N))) * 1e-3).astype(np.complex64)
    base = 5000
    # (mode, l, mid, r): wide and very narrow slices (a slice of one or two Inf bins gives +-Inf AND NaN outputs)
    specs = [("USB", base, float(base), base + 80), ("LSB", base + 200, base + 280.5, base + 280), ("USB", base + 400, base + 400.0, base + 401),
             ("LSB", base + 500, base + 502.0, base + 502), ("USB", base + 600, base + 601.5, base + 603), ("AM", base + 700, base + 750.0, base + 800),
             ("FM", base + 900, base + 950.5, base + 1000)]
    vals = [complex(np.inf, 0), complex(0, np.inf), complex(-np.inf, 0), complex(np.inf, np.inf), complex(np.nan, 0), complex(np.inf, -np.inf)]
    bad = {}
    for ci, (_, l, _, r) in enumerate(specs):
        frames = sorted(rng.choice(nframes - 6, size=14, replace=False))
        for k, f in enumerate(frames):
            b = int(rng.integers(l, r))
            spec[f, b] = vals[(k + ci) % len(vals)]
            if k % 5 == 0:  # a run of two bad frames
                spec[f + 1, b] = vals[(k + ci + 3) % len(vals)]
            bad.setdefault(ci, []).append(int(f))

    def run(F):
        ctx = Context(N, False, levels, additional_size=n, audio_fft_size=n, audio_rate=12000, input_format="s16", max_batch=F,
                      max_clients=len(specs))
        try:
            d = ctx.dev_alloc(spec.nbytes)
            ctx.h2d(d, spec)
            cl = []
            for mode, l, mid, r in specs:
                g = AudioClient(ctx)
                g.set_audio_demodulation(mode)
                g.set_audio_range(l, mid, r)
                cl.append(g)
            out = [[] for _ in specs]
            f = 0
            while f < nframes:
                nb = min(F, nframes - f)
                rc = ctx.lib.psdr_demod_batch_from(ctx.h, C.c_void_p(d.value + f * N * 8), N, nb, f)
                assert rc == 0, ctx.lib.psdr_last_error()
                for ci, g in enumerate(cl):
                    a, p, nan = g.read_audio(F)
                    out[ci].append((a[:nb].copy(), p[:nb].copy(), nan[:nb].copy()))
                f += nb
            ctx.dev_free(d)
            return [tuple(np.concatenate([b[k] for b in o]) for k in range(3)) for o in out]
        finally:
            ctx.close()

    want = run(1)
    dropped_ssb = sum(int(want[ci][2].sum()) for ci in range(5))
    assert dropped_ssb >= 40, dropped_ssb
    # the recurrence has something to say: a slot whose tail went bad drops finite frames too, another one recovers
    stuck = [ci for ci in range(5) if want[ci][2][-6:].all()]
    alive = [ci for ci in range(5) if not want[ci][2][-6:].any()]
    for F in (5, 8, 37):
        got = run(F)
        for ci in range(len(specs)):
            tag = f"n={n} chain={chain} F={F} client {ci} {specs[ci][0]} (bad frames {bad[ci]})"
            assert np.array_equal(got[ci][2], want[ci][2]), (tag, np.nonzero(got[ci][2] != want[ci][2])[0][:10])
            assert _eq_nan(got[ci][1], want[ci][1]), tag
            ok = want[ci][2] == 0  # (the audio of a dropped frame is not served; its row holds y + prev of whichever tail)
            assert _eq_nan(got[ci][0][ok], want[ci][0][ok]), (tag, np.nonzero((got[ci][0] != want[ci][0]).any(axis=1) & ok)[0][:10])
    assert stuck or alive  # (at least one of the two outcomes occurred; which one depends on the transform's arithmetic)


@pytest.mark.parametrize("n,chain", [(360, 1), (248, 1)])
def test_inf_input_samples_whole_path_batched_equals_frame_by_frame(n, chain, monkeypatch):
    """The same through the whole path: +-Inf SAMPLES in an f32 ring (every bin of the two frames that hold the half-frame
    becomes NaN or +-Inf), batches of 1, 6 and 19 frames: flags, pwr and every served frame's audio bit for bit."""
    from phantomsdr_amd import AudioClient, Context
    monkeypatch.setenv("PSDR_DEMOD_CHAIN", str(chain))
    monkeypatch.setenv("PSDR_DEMOD_K", "4")
    N, nframes = 1 << 14, 38
    levels = levels_for(N)
    x = synth_stream((nframes + 1) * (N // 2), False, seed=15, fft_size=N).astype(np.complex64)
    halves = x.reshape(nframes + 1, N // 2).copy()
    for h, k, v in [(3, 17, complex(np.inf, 0)), (9, 100, complex(0, -np.inf)), (14, 5, complex(np.inf, -np.inf)), (15, 9, complex(-np.inf, 0)),
                    (20, 1, complex(np.nan, 0)), (27, 4000, complex(np.inf, np.inf))]:
        halves[h, k] = v
    raw = halves.reshape(-1).view(np.float32).copy()
    am = int((0.11 * N - (N // 2 + 1)) % N)
    specs = [("USB", am, float(am), am + 80), ("LSB", am - 80, am + 0.5, am), ("USB", am + 300, am + 300.0, am + 302), ("LSB", am + 500, am + 501.0, am + 501),
             ("AM", am - 100, float(am), am + 100), ("FM", am - 100, am + 0.5, am + 100)]

    def run(F):
        ctx = Context(N, False, levels, additional_size=n, audio_fft_size=n, audio_rate=12000, input_format="f32", max_batch=F,
                      max_clients=len(specs))
        try:
            d = ctx.dev_alloc(raw.nbytes)
            ctx.h2d(d, raw)
            cl = []
            for mode, l, mid, r in specs:
                g = AudioClient(ctx)
                g.set_audio_demodulation(mode)
                g.set_audio_range(l, mid, r)
                cl.append(g)
            out = [[] for _ in specs]
            hb = ctx.half_frame_bytes()
            f = 0
            while f < nframes:
                nb = min(F, nframes - f)
                ctx.process_batch(d, nb, offset_bytes=f * hb)
                ctx.demod_batch(f)
                for ci, g in enumerate(cl):
                    a, p, nan = g.read_audio(F)
                    out[ci].append((a[:nb].copy(), p[:nb].copy(), nan[:nb].copy()))
                f += nb
            ctx.dev_free(d)
            return [tuple(np.concatenate([b[k] for b in o]) for k in range(3)) for o in out]
        finally:
            ctx.close()

    want = run(1)
    assert sum(int(w[2].sum()) for w in want) >= 40
    for F in (6, 19):
        got = run(F)
        for ci in range(len(specs)):
            tag = f"n={n} chain={chain} F={F} client {ci} {specs[ci][0]}"
            assert np.array_equal(got[ci][2], want[ci][2]), (tag, np.nonzero(got[ci][2] != want[ci][2])[0][:10])
            assert _eq_nan(got[ci][1], want[ci][1]), tag
            ok = want[ci][2] == 0
            assert _eq_nan(got[ci][0][ok], want[ci][0][ok]), tag
