"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the
same seeded inputs.  Tolerances (north_star): float spectra 1e-4 relative (observed
~1e-6); audio 1e-4 relative L2; int8 pyramid bit-exact against the quantiser applied to
the GPU's own spectrum, and >= 99.9 % identical (rest +-1) against the oracle's."""
import numpy as np
import pytest

from helpers import check_fm, pwr_tolerance, quantize_raw, rel_err, rel_l2, synth_stream
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SPEC_TOL = 1e-4
AUDIO_TOL = 1e-4


def levels_for(R, waterfall_size=1024):
    lv, cur = 0, R
    while cur >= waterfall_size:
        lv += 1
        cur //= 2
    return max(lv, 1)


def halves_f32(N, is_real, nhalves, seed):
    x = synth_stream(nhalves * (N // 2), is_real, seed, fft_size=N)
    x = x.astype(np.float32) if is_real else x.astype(np.complex64)
    return x.reshape(nhalves, N // 2)


def check_pyramid(q_gpu, spec_gpu_k, q_orc, N, is_real, levels):
    q_self = O.pyramid_from_spectrum(spec_gpu_k, N, is_real, levels)
    assert np.array_equal(q_gpu, q_self), (
        f"int8 pyramid differs from the reference quantiser applied to the GPU's own spectrum: "
        f"{(q_gpu != q_self).sum()} of {q_gpu.size}")
    d = np.abs(q_gpu.astype(np.int16) - q_orc.astype(np.int16))
    assert d.max() <= 1, f"pyramid differs from oracle by {d.max()} LSB"
    assert (d != 0).mean() <= 1e-3, f"pyramid mismatch rate {(d != 0).mean():.2e}"


@pytest.mark.parametrize("N,is_real", [
    (1 << 12, 0), (1 << 13, 0), (1 << 14, 0), (1 << 15, 0), (1 << 16, 0), (1 << 17, 0),
    (1 << 18, 0), (1 << 19, 0), (1 << 20, 0), (1 << 21, 0),
    (1 << 13, 1), (1 << 14, 1), (1 << 15, 1), (1 << 16, 1), (1 << 17, 1), (1 << 19, 1),
    (1 << 21, 1), (1 << 22, 1),
])
def test_fft_plugin_level1(N, is_real):
    """HipFFT driven exactly like the reference drives class FFT (src/fft.cpp:17-30,61-98)."""
    from phantomsdr_amd import HipFFT
    R = N // 2 if is_real else N
    levels = levels_for(R)
    A = 360
    fo = O.FFT(N, is_real, levels, 0, A)
    fg = HipFFT(N, 1, levels, 0)
    fg.set_output_additional_size(A)
    if is_real:
        fg.plan_r2c(0)
    else:
        fg.plan_c2c(HipFFT.FORWARD, 0)
    nfl = N // 2 * (1 if is_real else 2)
    bufs = [fg.malloc(nfl) for _ in range(3)]
    h = halves_f32(N, is_real, 3, seed=100 + int(np.log2(N)) + 50 * is_real)
    for b, hh in zip(bufs, h):
        b[:] = hh.view(np.float32)
    try:
        for f in range(2):
            a1, a2 = bufs[f], bufs[f + 1]
            if is_real:
                fg.load_real_input(a1, a2)
            else:
                fg.load_complex_input(a1, a2)
            fg.execute()
            fo.load(h[f], h[f + 1])
            fo.execute()
            Xg = fg.get_output_buffer().copy()
            Xo = fo.output().copy()
            nb = N // 2 if is_real else N + A   # real: bin N/2 is compared separately
            assert rel_err(Xg[:nb], Xo[:nb]) < SPEC_TOL
            assert rel_l2(Xg[:nb], Xo[:nb]) < 1e-5
            if is_real:   # unnormalised Nyquist bin (src/fft_impl.cpp:156-160 never visits it)
                assert abs(Xg[N // 2] - Xo[N // 2]) <= 1e-4 * np.abs(Xo[: N // 2]).max() * N
            check_pyramid(fg.get_quantized_buffer().copy(), Xg, fo.quantized().copy(), N, is_real, levels)
    finally:
        for b in bufs:
            fg.free(b)
        fg.close()


@pytest.mark.parametrize("is_real", [0, 1])
@pytest.mark.parametrize("fmt", ["u8", "s8", "u16", "s16", "f32", "f64"])
def test_process_batch_formats(fmt, is_real):
    """raw ring in HBM -> device-side convert (src/samplereader.cpp:29-40) -> F frames."""
    from phantomsdr_amd import Context
    N, F = 1 << 14, 5
    R = N // 2 if is_real else N
    levels = levels_for(R)
    sigma = 2.0 ** -5 if fmt in ("u8", "s8") else 2.0 ** -9
    x = synth_stream((F + 1) * (N // 2), is_real, seed=7 + is_real, sigma=sigma, fft_size=N)
    raw = quantize_raw(x, fmt, is_real)
    conv = O.convert(raw, fmt)
    halves = (conv if is_real else conv.view(np.complex64)).reshape(F + 1, N // 2)
    ctx = Context(N, is_real, levels, input_format=fmt, max_batch=F)
    try:
        assert ctx.half_frame_bytes() * (F + 1) == raw.nbytes
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        ctx.process_batch(d, F)
        fo = O.FFT(N, is_real, levels)
        for f in range(F):
            fo.load(halves[f], halves[f + 1])
            fo.execute()
            Xg = ctx.read_spectrum(f)
            nb = N // 2 if is_real else N
            assert rel_err(Xg[:nb], fo.output()[:nb]) < SPEC_TOL, f"frame {f}"
            check_pyramid(ctx.read_quantized(f), Xg, fo.quantized().copy(), N, is_real, levels)
        ctx.dev_free(d)
    finally:
        ctx.close()


def run_demod_case(N, is_real, n, clients, nbatches, F, seed, fmt="s16", mode_changes=None):
    """clients: list of (mode, l, mid, r).  Returns nothing; asserts parity frame by frame."""
    from phantomsdr_amd import AudioClient, Context
    R = N // 2 if is_real else N
    levels = levels_for(R)
    nframes = nbatches * F
    x = synth_stream((nframes + 1) * (N // 2), is_real, seed=seed, fft_size=N)
    raw = quantize_raw(x, fmt, is_real)
    conv = O.convert(raw, fmt)
    halves = (conv if is_real else conv.view(np.complex64)).reshape(nframes + 1, N // 2)
    ctx = Context(N, is_real, levels, additional_size=n, audio_fft_size=n, audio_rate=12000,
                  input_format=fmt, max_batch=F, max_clients=len(clients))
    try:
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        gcl, ocl = [], []
        for mode, l, mid, r in clients:
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            g.set_audio_range(l, mid, r)
            gcl.append(g)
            o = O.AudioClient(is_real, n, 12000, R)
            o.set_audio_demodulation(mode)
            o.set_audio_range(l, mid, r)
            ocl.append(o)
        fo = O.FFT(N, is_real, levels, 0, n)
        hb = ctx.half_frame_bytes()
        frame = 0
        for b in range(nbatches):
            if mode_changes and b in mode_changes:
                for ci, mode in mode_changes[b]:
                    gcl[ci].set_audio_demodulation(mode)
                    ocl[ci].set_audio_demodulation(mode)
            ctx.process_batch(d, F, offset_bytes=b * F * hb)
            ctx.demod_batch(frame)
            got = [g.read_audio(F) for g in gcl]
            for f in range(F):
                fo.load(halves[frame], halves[frame + 1])
                fo.execute()
                spec = fo.output().copy()
                for ci, o in enumerate(ocl):
                    a_o, p_o, _, dropped = o.send_audio(spec, frame, fft=fo)
                    a_g, p_g, nan_g = got[ci][0][f], got[ci][1][f], got[ci][2][f]
                    assert not dropped and nan_g == 0
                    tag = f"client {ci} {clients[ci]} frame {frame}"
                    assert abs(p_g - p_o) <= pwr_tolerance(p_o, o.fwd_scale), tag
                    scale = max(np.abs(a_o).max(), 1e-30)
                    if o.mode == O.FM:
                        # 1e-4 rad, conditioned by |B|max / |B[i]| (helpers.fm_tolerance)
                        check_fm(a_g, a_o, o.baseband()[: o.n // 2], o.bb_prev, tag, fwd_scale=max(o.fwd_scale, o.fwd_scale_prev))
                    else:
                        assert rel_l2(a_g, a_o) < AUDIO_TOL, f"{tag}: rel L2 {rel_l2(a_g, a_o):.2e}"
                        assert np.abs(a_g - a_o).max() <= 2e-4 * scale, tag
                frame += 1
        ctx.dev_free(d)
    finally:
        ctx.close()


def tone_bin(N, is_real, fnorm):
    """client-coordinate bin of a tone at normalised frequency fnorm (cycles/sample)."""
    if is_real:
        return fnorm * N
    return (fnorm * N - (N // 2 + 1)) % N


@pytest.mark.parametrize("N,is_real,n", [(1 << 14, 0, 248), (1 << 15, 1, 360), (1 << 16, 0, 248)])
def test_demod_all_modes(N, is_real, n):
    R = N // 2 if is_real else N
    am = int(tone_bin(N, is_real, 0.11))
    fm = int(tone_bin(N, is_real, 0.31 if is_real else -0.21))
    w3, w5 = n // 4, n // 2 - 2
    clients = [
        ("USB", am, am + 0.0, am + w3),
        ("USB", am + 1, am + 1.5, am + 1 + w3),            # odd mid, fractional
        ("LSB", am - w3, float(am), am),
        ("LSB", am - w3 + 1, am + 1.25, am + 1),
        ("AM", am - w5, float(am), am + w5),
        ("AM", am - w5 + 1, am + 1.0, am + 1 + w5),
        ("FM", fm - w5, float(fm), fm + w5),
        ("FM", fm - w5 + 1, fm + 1.75, fm + w5),
        ("USB", 0, 0.0, w3),                                 # lower edge of the spectrum
        ("LSB", R - 1 - w3, float(R - 1), R - 1),            # upper edge
        ("USB", 10, 5.0, 10 + w3),                           # mid left of the slice
        ("AM", 40, 300.0, 60),                               # mid far outside: empty copy
        ("USB", 100, 100.0, 100),                            # empty slice
        ("USB", 200, 200.0, 200 + n),                        # widest slice allowed
    ]
    if not is_real:
        dc = N // 2 - 1                                      # k = 0 sits at c = N/2 - 1
        clients += [("USB", dc - 10, dc - 10.0, dc + 50),    # slice crossing the k-space wrap
                    ("AM", dc - w5, float(dc), dc + w5)]
    run_demod_case(N, is_real, n, clients, nbatches=3, F=4, seed=11 + is_real)


def test_demod_mode_switch_keeps_state():
    """buffers are not reset on a mode change (src/signal.cpp:316-328 only resets the AGC):
    the SSB tail survives AM frames and vice versa."""
    N, n = 1 << 14, 248
    am = int(tone_bin(N, 0, 0.11))
    clients = [("USB", am, float(am), am + 60), ("AM", am - 100, float(am), am + 100)]
    changes = {1: [(0, "AM"), (1, "USB")], 2: [(0, "USB"), (1, "FM")], 3: [(0, "LSB"), (1, "AM")]}
    run_demod_case(N, 0, n, clients, nbatches=4, F=3, seed=21, mode_changes=changes)


def test_demod_single_frame_batches():
    """F = 1: the real-time shape (one frame per call), state carried every call."""
    N, n = 1 << 14, 248
    am = int(tone_bin(N, 0, 0.11))
    fm = int(tone_bin(N, 0, -0.21))
    clients = [("USB", am, float(am), am + 60), ("AM", am - 100, float(am), am + 100),
               ("FM", fm - 100, float(fm), fm + 100), ("LSB", am - 60, float(am), am)]
    run_demod_case(N, 0, n, clients, nbatches=6, F=1, seed=31)


@pytest.mark.parametrize("n", [8, 60, 124, 256, 720, 1000, 2048, 2 * 839 * 2, 10068])
def test_demod_audio_fft_sizes(n):
    """any multiple of 4, including large prime factors (31, 839) and power-of-two sizes."""
    N = 1 << 16
    am = int(tone_bin(N, 0, 0.11))
    w = min(n // 2 - 1, 400)
    clients = [("USB", am, float(am), am + min(n // 2, 200)), ("AM", am - w, float(am), am + w),
               ("LSB", am - min(n // 2, 200), float(am), am)]
    run_demod_case(N, 0, n, clients, nbatches=2, F=2, seed=41)


@pytest.mark.parametrize("n,is_real", [(360, 0), (720, 1), (720, 0)])
def test_demod_fixed_plans_all_modes(n, is_real):
    """the compile-time-plan inverse DFTs (n = 360: 8*9*5, n = 720: 8*9*10): every mode, odd and
    fractional mids (frame flips), slices at the spectrum edges, 5-frame batches (the last
    work-group of a launch is partly empty)."""
    N = 1 << 16
    R = N // 2 if is_real else N
    c = int(tone_bin(N, is_real, 0.13))
    w3, w5 = n // 4, n // 2 - 3
    clients = [("USB", c, float(c), c + w3), ("USB", c + 1, c + 1.5, c + 1 + w3), ("LSB", c - w3, float(c), c),
               ("LSB", c - w3 + 1, c + 1.25, c + 1), ("AM", c - w5, float(c), c + w5),
               ("AM", c - w5 + 1, c + 1.0, c + 1 + w5), ("FM", c - w5, float(c), c + w5),
               ("FM", c - w5 + 1, c + 1.75, c + w5), ("USB", 0, 0.0, w3), ("LSB", R - 1 - w3, float(R - 1), R - 1),
               ("AM", 40, 300.0, 60), ("USB", 200, 200.0, 200 + n), ("AM", c - 20, float(c), c + 31)]
    run_demod_case(N, is_real, n, clients, nbatches=2, F=5, seed=77 + n + is_real)


def test_demod_many_clients():
    """257 clients (mixed modes, random slices) x 3-frame batches: the demodulation kernels'
    grids are ragged in every dimension (items per work-group, clients per wave in the post
    chain's layout); every client of every frame against the oracle."""
    N, n = 1 << 14, 360
    rng = np.random.default_rng(123)
    clients = []
    for i in range(257):
        mode = ("USB", "LSB", "AM", "FM")[i % 4]
        m = int(rng.integers(200, N - 200))
        w = int(rng.integers(10, n // 2 - 2))
        l, r = (m, m + w) if mode == "USB" else (m - w, m) if mode == "LSB" else (m - w, m + w)
        clients.append((mode, l, m + float(rng.integers(0, 4)) * 0.25, r))
    run_demod_case(N, 0, n, clients, nbatches=2, F=3, seed=5)


def test_baseline_cfg1_shape():
    """BASELINE.json configs[0]: 3.2 MSPS IQ u8 (rtl_sdr), 2^16-point FFT, one audio client
    (n = ceil(12000 * 2^16 / 3.2e6 / 4) * 4 = 248) - the reference's own CPU-runnable case,
    frame by frame against the oracle from the raw u8 bytes on."""
    from phantomsdr_amd import derived_params
    p = derived_params(3_200_000, 1 << 16, False)
    assert p["audio_fft_size"] == 248 and p["fft_result_size"] == 1 << 16
    N = 1 << 16
    m = int(tone_bin(N, 0, 0.11))
    run_demod_case(N, 0, p["audio_fft_size"], [("USB", m, float(m), m + 61)], nbatches=3, F=4, seed=21,
                   fmt="u8")


def test_waterfall_batch():
    from phantomsdr_amd import Context, WaterfallClient
    N, is_real, F = 1 << 16, 0, 6
    R = N
    levels = levels_for(R)
    x = synth_stream((F + 1) * (N // 2), is_real, seed=5, fft_size=N)
    raw = quantize_raw(x, "s16", is_real)
    ctx = Context(N, is_real, levels, input_format="s16", max_batch=F, max_waterfall_clients=6,
                  skip_num=2)
    try:
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        w_full = WaterfallClient(ctx)                       # default: whole span, coarsest level
        w_zoom = WaterfallClient(ctx)
        assert w_zoom.on_window_message(1000, 1000 + 2048)  # picks the level by itself
        lv, l, r = O.waterfall_pick_level(levels, R >> (levels - 1), 1000, 1000 + 2048)
        assert (w_zoom.level, w_zoom.l, w_zoom.r) == (lv, l, r)
        w_l0 = WaterfallClient(ctx)
        w_l0.set_waterfall_range(0, 12345, 12345 + 1024)
        w_clamp = WaterfallClient(ctx)
        w_clamp.set_waterfall_range(levels - 1, 10, 10 ** 6)   # r is clamped to R >> level
        assert w_clamp.r == R >> (levels - 1)
        assert not w_full.on_window_message(5, 5)            # rejected (src/waterfall.cpp:56-58)
        first = 3                                            # frames 3..8: sent = 4, 6, 8
        ctx.process_batch(d, F)
        ctx.waterfall_batch(first)
        sent = [f for f in range(F) if (first + f) % 2 == 0]
        qs = [ctx.read_quantized(f) for f in range(F)]
        for w in (w_full, w_zoom, w_l0, w_clamp):
            got, label = w.read_waterfall()
            assert got.shape == (len(sent), w.r - w.l)
            assert label == (w.l << w.level, w.r << w.level)
            for si, f in enumerate(sent):
                assert np.array_equal(got[si], ctx.quantized_level(qs[f], w.level)[w.l:w.r])
        ctx.dev_free(d)
    finally:
        ctx.close()


@pytest.mark.parametrize("is_real", [0, 1])
def test_waterfall_many_clients(is_real):
    """64 waterfall clients (BASELINE.json configs[4]: zoomed waterfalls) at random levels and
    ranges, IQ (tile-major records for the low levels) and real input (level-major buffer): the
    gathered rows equal the corresponding slices of the frame's int8 pyramid."""
    from phantomsdr_amd import Context, WaterfallClient
    N, F = 1 << 16, 5
    R = N // 2 if is_real else N
    levels = levels_for(R)
    x = synth_stream((F + 1) * (N // 2), bool(is_real), seed=6, fft_size=N)
    raw = quantize_raw(x, "s16", bool(is_real))
    ctx = Context(N, is_real, levels, input_format="s16", max_batch=F, max_waterfall_clients=64, skip_num=1)
    try:
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        rng = np.random.default_rng(31)
        ws = []
        for i in range(64):
            w = WaterfallClient(ctx)
            lv = int(rng.integers(0, levels))
            span = R >> lv
            width = int(rng.integers(1, min(span, 3000) + 1))
            l = int(rng.integers(0, span - width + 1))
            w.set_waterfall_range(lv, l, l + width)
            ws.append(w)
        ctx.process_batch(d, F)
        ctx.waterfall_batch(0)
        qs = [ctx.read_quantized(f) for f in range(F)]
        for w in ws:
            got, label = w.read_waterfall()
            assert got.shape == (F, w.r - w.l)
            assert label == (w.l << w.level, w.r << w.level)
            for f in range(F):
                assert np.array_equal(got[f], ctx.quantized_level(qs[f], w.level)[w.l:w.r]), (w.level, w.l, w.r, f)
        ctx.dev_free(d)
    finally:
        ctx.close()


def test_error_paths():
    from phantomsdr_amd import AudioClient, Context, PsdrError
    with pytest.raises(PsdrError):
        Context(1000, 0, 1)                                  # not a power of two
    with pytest.raises(PsdrError):
        Context(1 << 10, 0, 1)                               # below the supported range
    ctx = Context(1 << 14, 0, 3, audio_fft_size=248, max_clients=2)
    try:
        a, b = AudioClient(ctx), AudioClient(ctx)
        with pytest.raises(PsdrError):
            AudioClient(ctx)                                 # slots exhausted
        assert not a.on_window_message(-1, 5.0, 10)          # src/signal.cpp:305-308
        assert not a.on_window_message(0, 5.0, 1 << 14)      # r >= fft_result_size
        assert not a.on_window_message(0, 5.0, 249)          # wider than audio_fft_size
        assert not a.on_window_message(20, 5.0, 10)
        assert not a.on_window_message(0, None, 10)          # no `m` in the message
        assert a.on_window_message(0, 5.0, 248)
        with pytest.raises(PsdrError):
            ctx.demod_batch(0)                               # nothing processed yet
        b.on_close()
        AudioClient(ctx)                                     # slot is reusable
    finally:
        ctx.close()


@pytest.mark.parametrize("audio_rate,F,nb,n,agc_form",
                         [(12000, 8, 5, 248, 1), (192000, 128, 5, 248, 1), (12000, 7, 6, 252, 1), (48000, 33, 5, 248, 1),
                          (6000, 9, 4, 360, 1), (44100, 40, 5, 248, 1),
                          (12000, 8, 5, 248, 0), (48000, 33, 5, 248, 0), (6000, 9, 4, 360, 0), (12000, 8, 6, 248, 2)])
def test_post_chain_bit_exact(audio_rate, F, nb, n, agc_form):
    """DC blocker + AGC + int16 conversion on the GPU (psdr_set_post_chain) against the oracle's
    chain fed with the SAME float audio (the GPU's own demodulator output): the recurrences are
    sequential f32, so the PCM must be identical.  Covers the AGC look-ahead start-up (2400
    samples at 12 kHz), several batches, a mode change (AGC reset, src/signal.cpp:316-328) and a
    client added late.  192000 is the audio_sps of the reference's shipped config.toml (WBFM): DC delay
    512 (k_pc_mad: the sums of the last 512 steps in an LDS ring), look-ahead 38400 samples (150 pieces per block; the batch
    is sized so that the look-ahead fills: 640 frames of 124 samples).  n = 252: frames of 126 samples - not whole row
    groups of the chain's lane-interleaved streams (the scalar gather / output kernels), 7 of them: streams that are
    not whole 16-step blocks.  48000 (the other shipped configs' rate: D = 128) and 6000 (D = 16): the two-wave kernel for
    any power-of-two delay, its ring of sums in LDS, streams that end inside a block (33 and 9 frames); 44100: D = 116,
    the generic two-kernel path with a division.  agc_form (psdr.h PSDR_OPT_POST_CHAIN_AGC): 1 = the default - maxima of
    16-sample chunks + ONE kernel for look-ahead peak, gain and int16 wherever the rate and the audio size allow it (all
    cases but n = 252: frames that are not whole row groups, and 44100: a look-ahead of 8820 samples is not whole chunks;
    the streams here end inside a chunk, inside a round of the kernel's pipeline, and - F = 8 / 9 - before its third round),
    0 = the five-kernel form everywhere, 2 = the form changes with every batch (psdr_set_option between batches: the chain's
    carried state - histories, running sums, gain - is the same in both)."""
    from phantomsdr_amd import AudioClient, Context
    N = 1 << 14
    R, levels = N, levels_for(N)
    nframes = nb * F
    x = synth_stream((nframes + 1) * (N // 2), False, seed=77, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    ctx = Context(N, False, levels, additional_size=n, audio_fft_size=n, audio_rate=audio_rate,
                  input_format="s16", max_batch=F, max_clients=4)
    try:
        ctx.set_option(ctx.OPT_POST_CHAIN_AGC, agc_form & 1)
        ctx.set_post_chain(True)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        specs = [("USB", 3000, 3010.0, 3200), ("AM", 5000, 5100.5, 5200), ("FM", 9000, 9100.0, 9200)]
        gcl, chains = [], []
        for mode, l, mid, r in specs:
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            g.set_audio_range(l, mid, r)
            gcl.append(g)
            chains.append(O.PostChain(audio_rate))
        hb = ctx.half_frame_bytes()
        total = 0
        for b in range(nb):
            if agc_form == 2:
                ctx.set_option(ctx.OPT_POST_CHAIN_AGC, b & 1)
            if b == 2:  # mode change of client 0: AGC reset
                gcl[0].set_audio_demodulation("LSB")
                chains[0].reset_agc()
            if b == 3:  # a client that joins late starts with fresh state
                g = AudioClient(ctx)
                g.set_audio_demodulation("USB")
                g.set_audio_range(12000, 12010.0, 12200)
                gcl.append(g)
                chains.append(O.PostChain(audio_rate))
            ctx.process_batch(d, F, offset_bytes=b * F * hb)
            ctx.demod_batch(b * F)
            for g, ch in zip(gcl, chains):
                audio, _, nan = g.read_audio(F)
                pcm = g.read_pcm(F)
                assert not nan.any()
                for f in range(F):
                    want = ch.process(audio[f])
                    assert np.array_equal(pcm[f], want), (
                        f"batch {b} frame {f}: {np.count_nonzero(pcm[f] != want)} of {want.size} samples differ, "
                        f"max |d| {np.abs(pcm[f] - want).max()}")
                    total += int(np.count_nonzero(want))
        assert total > 1000, "the AGC never opened: the test did not exercise the chain"
        ctx.dev_free(d)
    finally:
        ctx.close()


@pytest.mark.parametrize("max_clients", [70, 600])
def test_post_chain_many_clients(max_clients):
    """70 clients (more than the 64 lanes of one wave of the chain's client-per-lane kernels),
    n = 360 (h = 180 is not a multiple of the kernels' 32-sample blocks), three batches.  600 slots: more than eight
    groups of 64 - the recurrence kernels then use whole waves (64 slots per work-group instead of 32)."""
    from phantomsdr_amd import AudioClient, Context
    N, n, F, nb = 1 << 14, 360, 6, 3
    x = synth_stream((nb * F + 1) * (N // 2), False, seed=78, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    ctx = Context(N, False, levels_for(N), additional_size=n, audio_fft_size=n, audio_rate=12000,
                  input_format="s16", max_batch=F, max_clients=max_clients)
    try:
        ctx.set_post_chain(True)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        rng = np.random.default_rng(9)
        gcl, chains = [], []
        for i in range(70):
            mode = ("USB", "LSB", "AM", "FM")[i % 4]
            m = int(rng.integers(300, N - 300))
            l, r = (m, m + 100) if mode == "USB" else (m - 100, m) if mode == "LSB" else (m - 100, m + 100)
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            g.set_audio_range(l, float(m), r)
            gcl.append(g)
            chains.append(O.PostChain(12000))
        hb = ctx.half_frame_bytes()
        for b in range(nb):
            ctx.process_batch(d, F, offset_bytes=b * F * hb)
            ctx.demod_batch(b * F)
            for ci, (g, ch) in enumerate(zip(gcl, chains)):
                audio, _, nan = g.read_audio(F)
                pcm = g.read_pcm(F)
                assert not nan.any()
                for f in range(F):
                    want = ch.process(audio[f])
                    assert np.array_equal(pcm[f], want), f"client {ci} batch {b} frame {f}"
        ctx.dev_free(d)
    finally:
        ctx.close()


@pytest.mark.parametrize("agc_form", [1, 0])
def test_post_chain_skips_nan_frames(agc_form):
    """A frame whose audio contains a NaN is dropped by the reference before the chain
    (src/signal.cpp:266-271): the chain's state must advance only over the surviving frames.  A second client beside it
    loses none: the two lanes of the chain's kernels (lane = slot) walk streams of different lengths in the same wave.
    The PCM rows of the dropped frames are zero (either form of the AGC, psdr.h PSDR_OPT_POST_CHAIN_AGC)."""
    import ctypes as C
    from phantomsdr_amd import AudioClient, Context
    from phantomsdr_amd._lib import check
    N, n, F, nb = 1 << 14, 248, 8, 4
    levels = levels_for(N)
    x = synth_stream((nb * F + 1) * (N // 2), False, seed=78, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    ctx = Context(N, False, levels, additional_size=n, audio_fft_size=n, audio_rate=12000,
                  input_format="s16", max_batch=F, max_clients=3)
    try:
        ctx.set_option(ctx.OPT_POST_CHAIN_AGC, agc_form)
        ctx.set_post_chain(True)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        g = AudioClient(ctx)
        g.set_audio_demodulation("USB")
        g.set_audio_range(3000, 3010.0, 3200)
        ch = O.PostChain(12000)
        g2 = AudioClient(ctx)
        g2.set_audio_demodulation("AM")
        g2.set_audio_range(8000, 8100.0, 8200)
        ch2 = O.PostChain(12000)
        hb = ctx.half_frame_bytes()
        poison = np.full(4, np.nan, np.float32)
        dropped_total = 0
        for b in range(nb):
            ctx.process_batch(d, F, offset_bytes=b * F * hb)
            p, nbytes = C.c_void_p(), C.c_size_t()
            check(ctx.lib.psdr_spectrum_device_ptr(ctx.h, 0, C.byref(p), C.byref(nbytes)))
            for f in ((2, 5) if b % 2 == 0 else (0,)):  # NaN into two bins of the client's slice
                ctx.synchronize()
                ctx.h2d(p, poison, offset=(f * N + 3050) * 8)
            ctx.demod_batch(b * F)
            audio, _, nan = g.read_audio(F)
            pcm = g.read_pcm(F)
            for f in range(F):
                if nan[f]:
                    dropped_total += 1
                    assert not pcm[f].any(), f"batch {b} frame {f}: the PCM row of a dropped frame is zero"
                    continue
                want = ch.process(audio[f])
                assert np.array_equal(pcm[f], want), f"batch {b} frame {f}"
            audio2, _, nan2 = g2.read_audio(F)
            pcm2 = g2.read_pcm(F)
            assert not nan2.any()
            for f in range(F):
                assert np.array_equal(pcm2[f], ch2.process(audio2[f])), f"batch {b} frame {f}: the client beside the poisoned one"
        assert dropped_total >= 5
        ctx.dev_free(d)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,F", [(360, 7), (720, 9), (360, 1)])
def test_chain_demodulation_is_bit_identical_to_the_two_kernel_path(n, F, monkeypatch):
    """k_demod_chain_fixed (transform + overlap-add + AM / FM in one kernel, one wave per chain of K frames, the
    previous frame's tail in registers, warm-up transforms at chain starts) against k_demod_idft_fixed + k_demod_ola
    on the same samples: audio, power and NaN flags of USB / LSB / AM / FM clients over two batches, for chains of
    1, 2, 3 frames (every chain start inside the batch needs its warm-up; FM needs two frames of it) and the default."""
    from phantomsdr_amd import SpectrumEngine
    N, nb = 1 << 17, 2
    x = synth_stream((nb * F + 1) * (N // 2), False, seed=5, fft_size=N)
    raw = quantize_raw(x, "s16", False)

    def run(chain, k):
        monkeypatch.setenv("PSDR_DEMOD_CHAIN", "1" if chain else "0")
        if k:
            monkeypatch.setenv("PSDR_DEMOD_K", str(k))
        else:
            monkeypatch.delenv("PSDR_DEMOD_K", raising=False)
        eng = SpectrumEngine(4369067, N, False, input_format="s16", max_batch=F, max_clients=8,
                             audio_sps=12000 if n == 360 else 24000)
        try:
            assert eng.params["audio_fft_size"] == n
            cl = [eng.add_audio_client(1000 + 3000 * i, 1000 + 3000 * i + (0 if m == "USB" else 120), 1000 + 3000 * i + 240, m)
                  for i, m in enumerate(["USB", "LSB", "AM", "FM", "FM", "AM"])]
            d = eng.ctx.dev_alloc(raw.nbytes)
            eng.ctx.h2d(d, raw)
            hb = eng.ctx.half_frame_bytes()
            out = []
            import ctypes as C
            from phantomsdr_amd._lib import check
            poison = np.full(2, np.nan, np.float32)
            for b in range(nb):
                eng.ctx.process_batch(d, F, offset_bytes=b * F * hb)
                if b == 0 and F > 2:
                    # NaN into one bin of the LSB and the FM client's windows, in a frame inside a chain and in the
                    # batch's last frame (its tail is the next batch's carried state): the NaN guard's flags
                    # (src/signal.cpp:266-271) and the poisoned overlap of the frame after must come out the same
                    eng.ctx.synchronize()
                    for f in (2, F - 1):
                        for bin_ in (1000 + 3000 * 1 + 50, 1000 + 3000 * 3 + 130):
                            p, nbytes = C.c_void_p(), C.c_size_t()
                            check(eng.ctx.lib.psdr_spectrum_device_ptr(eng.ctx.h, f, C.byref(p), C.byref(nbytes)))
                            eng.ctx.h2d(p, poison, offset=bin_ * 8)  # (2^17 points: the device keeps client order as it is)
                eng.ctx.demod_batch(b * F)
                eng.ctx.synchronize()
                out.append([c.read_audio(F) for c in cl])
            eng.ctx.dev_free(d)
            return out
        finally:
            eng.close()
    ref = run(False, 0)
    assert np.abs(np.nan_to_num(np.asarray(ref[1][3][0]))).max() > 0
    if F > 2:
        assert ref[0][1][2][2] == 1 and ref[0][3][2][F - 1] == 1 and ref[0][0][2].sum() == 0  # flags where the NaNs went
    for k in (0, 1, 2, 3):
        got = run(True, k)
        for b in range(nb):
            for ci in range(6):
                for name, u, v in zip(("audio", "pwr", "nan"), ref[b][ci], got[b][ci]):
                    u, v = np.asarray(u), np.asarray(v)
                    if u.dtype == np.float32:  # NaNs in the same places (payloads may differ), every other value bit for bit
                        nu, nv = np.isnan(u), np.isnan(v)
                        same = np.array_equal(nu, nv) and np.array_equal(u[~nu].view(np.uint32), v[~nv].view(np.uint32))
                    else:
                        same = np.array_equal(u, v)
                    assert same, f"K={k} batch {b} client {ci} {name}"


@pytest.mark.gpu
@pytest.mark.parametrize("N,F,nb,sps,every", [(1 << 17, 48, 10, 4369067, 3), (1 << 20, 128, 5, 34952534, 3),
                                                (1 << 17, 48, 11, 4369067, 99), (1 << 20, 128, 8, 34952534, 99)])
def test_post_chain_pipeline_matches_the_drained_sequence(N, F, nb, sps, every):
    """The post chain runs as a pipeline across batches: index + gather behind the demodulation, the moving averages on one
    stream, peak / gain / int16 on another, all beside the next FFT passes; what the stages hand on rotates over THREE
    buffer sets, the histories are copied into the next set, events order the reuse of a set three batches later.  The
    same batches with a full synchronisation after every call cannot overlap anything: PCM, audio and NaN flags of every
    client must be the same bits in both schedules.  every = 3: the piped run reads back (and drains) every third batch;
    every = 99: only at the very end - batch b + 3 takes batch b's set with batch b's chain possibly still running."""
    from phantomsdr_amd import SpectrumEngine
    if N <= 1 << 17:
        x = synth_stream((nb * F + 1) * (N // 2), False, seed=11, fft_size=N)
        raw = quantize_raw(x, "s16", False)
    else:  # (the bench's launch shape: passes long enough for every stage to overlap them; plain noise is enough here)
        raw = np.random.default_rng(11).integers(-3000, 3000, size=(nb * F + 1) * N, dtype=np.int16)

    def run(drained):
        eng = SpectrumEngine(sps, N, False, input_format="s16", max_batch=F, max_clients=12, audio_sps=12000)
        try:
            assert eng.params["audio_fft_size"] == 360
            # (the drained reference in the five-kernel form of the AGC, the piped run in the default one: chunk maxima + one
            # kernel - two schedules AND two forms, one set of bits)
            eng.ctx.set_option(eng.ctx.OPT_POST_CHAIN_AGC, 0 if drained else 1)
            eng.ctx.set_post_chain(True)
            cl = [eng.add_audio_client(1000 + 3000 * i, 1000 + 3000 * i + (0 if m == "USB" else 120), 1000 + 3000 * i + 240, m)
                  for i, m in enumerate(["USB", "LSB", "AM", "FM"] * 3)]
            d = eng.ctx.dev_alloc(raw.nbytes)
            eng.ctx.h2d(d, raw)
            hb = eng.ctx.half_frame_bytes()
            out = []
            for b in range(nb):
                eng.ctx.process_batch(d, F, offset_bytes=b * F * hb)
                if drained:
                    eng.ctx.synchronize()
                eng.ctx.demod_batch(b * F)
                if drained:
                    eng.ctx.synchronize()
                if drained or b % every == every - 1 or b == nb - 1:  # the piped run reads back only now and then
                    out.append((b, [(c.read_pcm(F).copy(),) + tuple(np.asarray(v).copy() for v in c.read_audio(F)) for c in cl]))
            eng.ctx.dev_free(d)
            return dict(out)
        finally:
            eng.close()
    ref, got = run(True), run(False)
    assert len(got) >= (2 if every < nb else 1)
    for b, clients in got.items():
        for ci, (pcm, audio, pwr, nan) in enumerate(clients):
            rp, ra, rw, rn = ref[b][ci]
            assert np.array_equal(pcm, rp), f"batch {b} client {ci} pcm"
            assert np.array_equal(audio.view(np.uint32), ra.view(np.uint32)) and np.array_equal(nan, rn)
    assert any(np.abs(c[0]).max() > 0 for c in got[nb - 1])


@pytest.mark.gpu
@pytest.mark.parametrize("nclients,F,n", [(40, 7, 360), (90, 5, 248), (5, 33, 360)])
def test_post_chain_agc_forms_agree_under_churn(nclients, F, n):
    """The two forms of the post chain's AGC (psdr.h PSDR_OPT_POST_CHAIN_AGC: chunk maxima + one four-wave kernel / the five
    kernels) on the same batches with everything that makes the lanes of a work-group differ: clients that join late (a
    look-ahead buffer still filling), pause for a batch (an empty stream, state frozen), change mode (AGC reset), leave (the
    slot re-used by a new client), frames dropped by the NaN guard (streams of different lengths inside one wave: 7 frames of
    180 samples end inside a 16-sample chunk), 40 / 90 clients = more than one work-group of 32 slots.  PCM bit for bit
    between the forms in every batch, and against the oracle's chain for the clients that are never paused or poisoned."""
    import ctypes as C
    from phantomsdr_amd import AudioClient, Context
    from phantomsdr_amd._lib import check
    N, nb = 1 << 14, 7
    levels = levels_for(N)
    x = synth_stream((nb * F + 1) * (N // 2), False, seed=91, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    modes = ("USB", "LSB", "AM", "FM")

    def run(form):
        rng = np.random.default_rng(5)
        ctx = Context(N, False, levels, additional_size=n, audio_fft_size=n, audio_rate=12000, input_format="s16", max_batch=F,
                      max_clients=nclients + 2)
        out, audio_out = [], []
        try:
            ctx.set_option(ctx.OPT_POST_CHAIN_AGC, form)
            ctx.set_post_chain(True)
            d = ctx.dev_alloc(raw.nbytes)
            ctx.h2d(d, raw)
            hb = ctx.half_frame_bytes()

            def add(i):
                g = AudioClient(ctx)
                mode = modes[i % 4]
                m = int(rng.integers(400, N - 400))
                l, r = (m, m + 100) if mode == "USB" else (m - 100, m) if mode == "LSB" else (m - 100, m + 100)
                g.set_audio_demodulation(mode)
                g.set_audio_range(l, float(m), r)
                g.centre = m
                return g
            cl = [add(i) for i in range(nclients - nclients // 4)]  # a quarter joins later
            for b in range(nb):
                if b == 2:
                    cl += [add(100 + i) for i in range(nclients // 4)]
                if b == 3:
                    for g in cl[1::5]:
                        g.set_paused(True)
                    cl[2].set_audio_demodulation("AM" if cl[2].demodulation < 2 else "USB")
                if b == 4:
                    for g in cl[1::5]:
                        g.set_paused(False)
                    cl[3].on_close()
                    cl[3] = add(200)
                ctx.process_batch(d, F, offset_bytes=b * F * hb)
                if b in (1, 5):  # NaN into the slice of client 0 in two frames: dropped by the NaN guard
                    p, nbytes = C.c_void_p(), C.c_size_t()
                    check(ctx.lib.psdr_spectrum_device_ptr(ctx.h, 0, C.byref(p), C.byref(nbytes)))
                    ctx.synchronize()
                    for f in (1, F - 1):
                        k = cl[0].centre + 20  # (client 0 is USB: its slice is [centre, centre + 100))
                        ctx.h2d(p, np.full(2, np.nan, np.float32), offset=(f * N + k) * 8)
                ctx.demod_batch(b * F)
                sits_out = set(range(1, len(cl), 5)) if b == 3 else set()  # (a paused client has no results to read in that batch)
                out.append([None if i in sits_out else g.read_pcm(F).copy() for i, g in enumerate(cl)])
                audio_out.append([None if i in sits_out else tuple(np.asarray(v).copy() for v in g.read_audio(F)) for i, g in enumerate(cl)])
            ctx.dev_free(d)
        finally:
            ctx.close()
        return out, audio_out
    (one, audio), (five, _) = run(1), run(0)
    dropped = 0
    for b in range(nb):
        assert len(one[b]) == len(five[b])
        for ci, (p1, p5) in enumerate(zip(one[b], five[b])):
            if p1 is None:
                continue
            assert np.array_equal(p1, p5), f"batch {b} client {ci}: {np.count_nonzero(p1 != p5)} samples differ between the forms"
        dropped += int(audio[b][0][2].sum())
    assert dropped >= 2, "the NaN guard dropped nothing: the test did not exercise streams of different lengths"
    # against the oracle: clients that are never paused, poisoned, re-moded or replaced (indices 5.. that are not 1 mod 5)
    for ci in [i for i in range(5, len(one[0])) if i % 5 != 1][:6]:
        ch = O.PostChain(12000)
        opened = 0
        for b in range(nb):
            a, _, nan = audio[b][ci]
            assert not nan.any()
            for f in range(F):
                want = ch.process(a[f])
                assert np.array_equal(one[b][ci][f], want), f"client {ci} batch {b} frame {f} against the oracle"
                opened += int(np.count_nonzero(want))
        assert opened > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,agc_form", [(248, 1), (248, 0), (252, 1), (360, 1)])
def test_post_chain_pcm16_rows_are_the_int32_rows(n, agc_form):
    """PSDR_OPT_POST_CHAIN_PCM16 = 1: the chain's output kernels (k_pc_agc, k_pc_out4 - and k_pc_out where frames are not whole
    row groups: n = 252 -, k_pc_zero for dropped frames) store int16 rows; psdr_read_pcm still delivers the reference's int32
    buffer (widened on the host), psdr_fetch_begin(PSDR_FETCH_PCM) moves half the bytes and psdr_fetched_pcm16 hands the rows
    out (psdr_fetched_audio's pcm is NULL then).  Against the oracle's chain on the GPU's own float audio, bit for bit; then
    the option off again: int32 rows as before."""
    import ctypes as C
    from phantomsdr_amd import AudioClient, Context
    from phantomsdr_amd._lib import check
    N, F, nb = 1 << 14, 24, 4
    x = synth_stream((nb * F + 1) * (N // 2), False, seed=79, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    ctx = Context(N, False, levels_for(N), additional_size=n, audio_fft_size=n, audio_rate=12000, input_format="s16", max_batch=F,
                  max_clients=6)
    try:
        ctx.set_option(ctx.OPT_POST_CHAIN_AGC, agc_form)
        ctx.set_option(ctx.OPT_POST_CHAIN_PCM16, 1)
        ctx.set_post_chain(True)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        gcl, chains = [], []
        for i, mode in enumerate(("USB", "AM", "FM", "LSB")):
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            m = 2000 + 2500 * i
            l, r = (m, m + 100) if mode == "USB" else (m - 100, m) if mode == "LSB" else (m - 100, m + 100)
            g.set_audio_range(l, float(m), r)
            gcl.append(g)
            chains.append(O.PostChain(12000))
        hb = ctx.half_frame_bytes()
        opened = 0
        for b in range(nb):
            if b == nb - 1:
                ctx.set_option(ctx.OPT_POST_CHAIN_PCM16, 0)  # (drains; the state of the chain carries over)
            ctx.process_batch(d, F, offset_bytes=b * F * hb)
            if b == 1:  # a frame of client 0 dropped by the NaN guard: its row is zero in either width
                p, nbytes = C.c_void_p(), C.c_size_t()
                check(ctx.lib.psdr_spectrum_device_ptr(ctx.h, 0, C.byref(p), C.byref(nbytes)))
                ctx.synchronize()
                ctx.h2d(p, np.full(2, np.nan, np.float32), offset=(3 * N + 2050) * 8)
            ctx.demod_batch(b * F)
            ctx.fetch_begin(ctx.FETCH_PCM)
            ctx.fetch_end()
            for ci, (g, ch) in enumerate(zip(gcl, chains)):
                audio, _, nan = g.read_audio(F)
                pcm = g.read_pcm(F)
                for f in range(F):
                    _, _, _, p32 = ctx.fetched_audio(g.id, f, pcm=True)
                    if b < nb - 1:
                        assert p32 is None
                        row = ctx.fetched_pcm16(g.id, f)
                        assert row.dtype == np.int16 and np.array_equal(row.astype(np.int32), pcm[f]), (b, ci, f)
                    else:
                        assert np.array_equal(p32, pcm[f]), (b, ci, f)
                    if nan[f]:
                        assert not pcm[f].any()
                        continue
                    want = ch.process(audio[f])
                    assert np.array_equal(pcm[f], want), f"batch {b} client {ci} frame {f}"
                    opened += int(np.count_nonzero(want))
        assert opened > 1000
        ctx.dev_free(d)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("agc_form", [1, 0])
def test_post_chain_batches_shorter_than_max_batch(agc_form):
    """Batches of 7, 16, 3, 16, 1, 12 frames in a context sized for 16: the chain's streams are the frames that WERE processed
    (the moving averages read them from the demodulator's rows where they lie - rows [slot][max_batch][h], a batch fills the
    first nframes of them), its histories carry over whatever the batch lengths; 124-sample frames, so most streams end
    inside a 16-sample chunk.  Against the oracle's chain, bit for bit."""
    from phantomsdr_amd import AudioClient, Context
    N, n, MB = 1 << 14, 248, 16
    sizes = [7, 16, 3, 16, 1, 12, 16, 16, 5]
    total = sum(sizes)
    x = synth_stream((total + 1) * (N // 2), False, seed=80, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    ctx = Context(N, False, levels_for(N), additional_size=n, audio_fft_size=n, audio_rate=12000, input_format="s16", max_batch=MB,
                  max_clients=5)
    try:
        ctx.set_option(ctx.OPT_POST_CHAIN_AGC, agc_form)
        ctx.set_post_chain(True)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        gcl, chains = [], []
        for i, mode in enumerate(("USB", "AM", "FM", "LSB", "AM")):
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            m = 1500 + 2900 * i
            l, r = (m, m + 100) if mode == "USB" else (m - 100, m) if mode == "LSB" else (m - 100, m + 100)
            g.set_audio_range(l, float(m), r)
            gcl.append(g)
            chains.append(O.PostChain(12000))
        hb = ctx.half_frame_bytes()
        f0 = opened = 0
        for F in sizes:
            ctx.process_batch(d, F, offset_bytes=f0 * hb)
            ctx.demod_batch(f0)
            for ci, (g, ch) in enumerate(zip(gcl, chains)):
                audio, _, nan = g.read_audio(MB)
                pcm = g.read_pcm(MB)
                assert not nan[:F].any()
                for f in range(F):
                    want = ch.process(audio[f])
                    assert np.array_equal(pcm[f], want), f"batch at frame {f0} ({F} frames) client {ci} frame {f}"
                    opened += int(np.count_nonzero(want))
            f0 += F
        assert opened > 1000
        ctx.dev_free(d)
    finally:
        ctx.close()
