"""Size-independent properties of the HIP path at BASELINE.json's full sizes:
the oracle takes minutes for 64 frames of 2^20 points, so the big-batch machinery (persistent
work-groups, tile tickets, double-buffered result sets, state carried across batches) is
checked through invariants instead: splitting a stream into batches must not change a bit,
Parseval, linearity, and a tone must land in its client-order bin."""
import numpy as np
import pytest

from helpers import quantize_raw, synth_stream

pytestmark = pytest.mark.gpu


def levels_for(R, waterfall_size=1024):
    lv, cur = 0, R
    while cur >= waterfall_size:
        lv += 1
        cur //= 2
    return max(lv, 1)


def _run(N, is_real, raw, splits, clients, n=360):
    """process the same raw stream with the given batch splits; returns per-frame spectra
    checksums, pyramids and the clients' audio."""
    from phantomsdr_amd import AudioClient, Context
    R = N // 2 if is_real else N
    F = max(splits)
    ctx = Context(N, is_real, levels_for(R), additional_size=n, audio_fft_size=n, audio_rate=12000,
                  input_format="s16", max_batch=F, max_clients=max(len(clients), 1))
    try:
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        gcl = []
        for mode, l, mid, r in clients:
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            g.set_audio_range(l, mid, r)
            gcl.append(g)
        hb = ctx.half_frame_bytes()
        spec, pyr, audio = [], [], [[] for _ in clients]
        frame = 0
        for nf in splits:
            ctx.process_batch(d, nf, offset_bytes=frame * hb)
            ctx.demod_batch(frame)
            for ci, g in enumerate(gcl):
                a, p, nan = g.read_audio(nf)
                assert not nan.any()
                audio[ci].append(a.copy())
            for f in range(nf):
                spec.append(ctx.read_spectrum(f).copy())
                pyr.append(ctx.read_quantized(f).copy())
            frame += nf
        ctx.dev_free(d)
        return spec, pyr, [np.concatenate(a) for a in audio]
    finally:
        ctx.close()


@pytest.mark.parametrize("N,is_real", [(1 << 20, 0), (1 << 21, 1)])
def test_batch_split_is_bit_invariant(N, is_real):
    """one batch of 24 frames == the same stream as 1+7+16 frames: spectrum, int8 pyramid and
    every client's audio, bit for bit (cfg2 / cfg3 shapes)."""
    R = N // 2 if is_real else N
    nframes = 24
    x = synth_stream((nframes + 1) * (N // 2), bool(is_real), seed=5, fft_size=N)
    raw = quantize_raw(x, "s16", bool(is_real))
    rng = np.random.default_rng(11)
    clients = []
    for i, mode in enumerate(["USB", "LSB", "AM", "FM", "USB", "AM"]):
        m = int(rng.uniform(0.1 * R, 0.9 * R))
        w = 90
        l, r = (m, m + w) if mode == "USB" else (m - w, m) if mode == "LSB" else (m - w, m + w)
        clients.append((mode, l, float(m) + 0.25 * (i % 3), r))
    a = _run(N, is_real, raw, [24], clients)
    b = _run(N, is_real, raw, [1, 7, 16], clients)
    for f in range(nframes):
        assert np.array_equal(a[0][f].view(np.uint32), b[0][f].view(np.uint32)), f"spectrum of frame {f}"
        assert np.array_equal(a[1][f], b[1][f]), f"pyramid of frame {f}"
    for ci in range(len(clients)):
        assert np.array_equal(a[2][ci].view(np.uint32), b[2][ci].view(np.uint32)), f"audio of client {ci}"


def test_bench_batch_size_matches_small_batches():
    """the bench's batch size (256 frames per launch at 2^20 points) gives the same bits as four
    launches of 64: per-frame CRCs of spectrum and pyramid, and the clients' audio."""
    import zlib
    from phantomsdr_amd import AudioClient, Context
    N, n, nframes = 1 << 20, 360, 256
    x = synth_stream((nframes + 1) * (N // 2), False, seed=21, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    rng = np.random.default_rng(5)
    clients = []
    for mode in ["USB", "LSB", "AM", "FM"]:
        m = int(rng.uniform(0.1 * N, 0.9 * N))
        l, r = (m, m + 90) if mode == "USB" else (m - 90, m) if mode == "LSB" else (m - 90, m + 90)
        clients.append((mode, l, float(m) + 0.5, r))

    def run(splits):
        ctx = Context(N, False, levels_for(N), additional_size=n, audio_fft_size=n, audio_rate=12000,
                      input_format="s16", max_batch=max(splits), max_clients=len(clients))
        try:
            d = ctx.dev_alloc(raw.nbytes)
            ctx.h2d(d, raw)
            gcl = []
            for mode, l, mid, r in clients:
                g = AudioClient(ctx)
                g.set_audio_demodulation(mode)
                g.set_audio_range(l, mid, r)
                gcl.append(g)
            hb = ctx.half_frame_bytes()
            crcs, audio, frame = [], [[] for _ in clients], 0
            for nf in splits:
                ctx.process_batch(d, nf, offset_bytes=frame * hb)
                ctx.demod_batch(frame)
                for ci, g in enumerate(gcl):
                    audio[ci].append(g.read_audio(nf)[0].copy())
                for f in range(0, nf, 5):  # every fifth frame: 8 MiB of spectrum each
                    crcs.append((zlib.crc32(ctx.read_spectrum(f).tobytes()), zlib.crc32(ctx.read_quantized(f).tobytes())))
                frame += nf
            ctx.dev_free(d)
            return crcs, [np.concatenate(a) for a in audio]
        finally:
            ctx.close()

    a = run([256])
    b = run([60, 60, 60, 60, 16])  # multiples of 5 keep the sampled frames aligned
    assert a[0] == b[0]
    for ci in range(len(clients)):
        assert np.array_equal(a[1][ci].view(np.uint32), b[1][ci].view(np.uint32)), f"audio of client {ci}"


def test_parseval_linearity_and_tone_bin_at_2_20():
    from phantomsdr_amd import Context
    N = 1 << 20
    rng = np.random.default_rng(3)
    k0 = 123457  # tone at reference bin k0 -> client-order index (k0 - N/2 - 1) mod N
    t = np.arange(N + N // 2)
    tone = 0.25 * np.exp(2j * np.pi * k0 * t / N)
    noise = (rng.standard_normal(t.size) + 1j * rng.standard_normal(t.size)) * 2.0 ** -6

    def frames(sig):
        s = np.stack([sig.real, sig.imag], -1).astype(np.float32).reshape(-1)
        ctx = Context(N, False, levels_for(N), input_format="f32", max_batch=2)
        try:
            d = ctx.dev_alloc(s.nbytes)
            ctx.h2d(d, s)
            ctx.process_batch(d, 2)
            return [ctx.read_spectrum(f).astype(np.complex128) for f in range(2)]
        finally:
            ctx.close()

    Xa, Xb, Xab = frames(tone), frames(noise), frames(tone + 0.5 * noise)
    n = np.arange(N)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / N)
    for f in range(2):
        # linearity
        lin = Xa[f] + 0.5 * Xb[f]
        assert np.abs(Xab[f] - lin).max() <= 2e-6 * np.abs(lin).max()
        # Parseval for the windowed, 1/N-normalised transform: sum |X|^2 = sum |w x|^2 / N
        seg = noise[f * (N // 2): f * (N // 2) + N]
        lhs = np.sum(np.abs(Xb[f]) ** 2)
        rhs = np.sum(np.abs((seg.real.astype(np.float32) + 1j * seg.imag.astype(np.float32)) * w) ** 2) / N
        assert abs(lhs - rhs) <= 1e-5 * rhs
        # the tone sits in reference bin k0 (k order), Hann main lobe: amplitude 0.25 * 0.5
        k = int(np.argmax(np.abs(Xa[f])))
        assert k == k0
        assert abs(abs(Xa[f][k]) - 0.125) < 1e-5
