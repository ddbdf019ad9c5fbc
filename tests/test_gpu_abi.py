"""GPU tests of the C-ABI's contracts that round-1 review found loose and of the
streaming-ingest ring: buffer capacities, the waterfall window of a batch, a non-power-of-two
waterfall_size (src/spectrumserver.cpp:56, src/waterfall.cpp:62-79), psdr_ring_*."""
import ctypes as C

import numpy as np
import pytest

from helpers import quantize_raw, synth_stream
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _levels(R, ws=1024):
    lv, cur = 0, R
    while cur >= ws:
        lv += 1
        cur //= 2
    return max(lv, 1)


def test_read_audio_refuses_short_buffers():
    """psdr_read_audio / psdr_read_pcm write exactly the frames of the last demod batch and refuse a
    buffer that holds fewer (round 1 wrote last_demod_frames rows into whatever it was given)."""
    from phantomsdr_amd import AudioClient, Context
    from phantomsdr_amd._lib import PsdrError
    N, F, n = 1 << 14, 6, 248
    ctx = Context(N, False, _levels(N), additional_size=n, audio_fft_size=n, input_format="s16", max_batch=F, max_clients=2)
    try:
        raw = quantize_raw(synth_stream((F + 1) * (N // 2), False, seed=3, fft_size=N), "s16", False)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        g = AudioClient(ctx)
        g.set_audio_range(100, 100.0, 160)
        ctx.process_batch(d, F)
        ctx.demod_batch(0)
        a, p, nan = g.read_audio()
        assert a.shape == (F, n // 2)
        big = g.read_audio(F + 3)                       # a larger buffer is fine: F rows come back
        assert big[0].shape == (F, n // 2) and np.array_equal(big[0], a)
        with pytest.raises(PsdrError) as e:
            g.read_audio(F - 1)
        assert e.value.code == -1
        # a smaller process_batch without a demod does not change what read_audio returns
        ctx.process_batch(d, 2)
        a2, _, _ = g.read_audio()
        assert a2.shape == (F, n // 2) and np.array_equal(a2, a)
        ctx.set_post_chain(True)
        ctx.process_batch(d, F)
        ctx.demod_batch(F)
        assert g.read_pcm().shape == (F, n // 2)
        with pytest.raises(PsdrError):
            g.read_pcm(F - 2)
        ctx.dev_free(d)
    finally:
        ctx.close()


def test_read_waterfall_uses_the_window_of_the_batch():
    """a window message between psdr_waterfall_batch and psdr_read_waterfall (another thread in the
    server) must not change the row length or labels of what was gathered"""
    from phantomsdr_amd import Context, WaterfallClient
    N, F = 1 << 16, 4
    levels = _levels(N)
    ctx = Context(N, False, levels, input_format="s16", max_batch=F, max_waterfall_clients=2, skip_num=1)
    try:
        raw = quantize_raw(synth_stream((F + 1) * (N // 2), False, seed=5, fft_size=N), "s16", False)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        w = WaterfallClient(ctx)
        w.set_waterfall_range(2, 1000, 1000 + 700)
        ctx.process_batch(d, F)
        ctx.waterfall_batch(0)
        qs = [ctx.read_quantized(f) for f in range(F)]
        w.set_waterfall_range(0, 5, 5 + 4096)            # the window moves AFTER the batch
        rows, label = w.read_waterfall()
        assert rows.shape == (F, 700) and label == (1000 << 2, 1700 << 2)
        for f in range(F):
            assert np.array_equal(rows[f], ctx.quantized_level(qs[f], 2)[1000:1700])
        ctx.waterfall_batch(0)                           # the next batch uses the new window
        rows, label = w.read_waterfall()
        assert rows.shape == (F, 4096) and label == (5, 5 + 4096)
        ctx.dev_free(d)
    finally:
        ctx.close()


@pytest.mark.parametrize("ws", [1000, 1024, 1500, 2048])
def test_waterfall_size_is_min_waterfall_fft(ws):
    """input.waterfall_size (src/spectrumserver.cpp:56) is the default width of a new client
    (src/websocket.cpp:198) and the target of the level search (src/waterfall.cpp:62-79), also when
    it is not a power of two (round 1 substituted R >> (levels-1))."""
    from phantomsdr_amd import Context, WaterfallClient
    N = 1 << 16
    levels = _levels(N, ws)
    ctx = Context(N, False, levels, input_format="s16", max_batch=1, max_waterfall_clients=4, waterfall_size=ws)
    try:
        w = WaterfallClient(ctx)
        assert (w.level, w.l, w.r) == (levels - 1, 0, min(ws, N >> (levels - 1)))
        rng = np.random.default_rng(ws)
        for _ in range(200):
            l = int(rng.integers(0, N - 2))
            r = int(rng.integers(l + 1, min(N, l + 1 + int(rng.integers(1, N)))))
            lv, ol, orr = O.waterfall_pick_level(levels, ws, l, r)
            assert w.on_window_message(l, r)
            assert (w.level, w.l) == (lv, ol) and w.r == min(orr, N >> lv), (l, r)
    finally:
        ctx.close()


@pytest.mark.parametrize("N,is_real", [(1 << 16, 0), (1 << 21, 1)])
def test_ring_ingest_matches_flat_upload(N, is_real):
    """psdr_ring_write_async + psdr_process_ring (pinned host halves -> HBM ring on a copy stream, writes
    running ahead of the transforms, the ring wrapping several times) give bit-identical spectra, int8
    pyramids and audio to psdr_process_batch over a flat upload of the same stream."""
    from phantomsdr_amd import AudioClient, Context
    from phantomsdr_amd._lib import PsdrError
    R = N // 2 if is_real else N
    n, F, nb, nh = 360, 4, 7, 8                     # 7 batches of 4 frames through an 8-half ring
    total = nb * F
    raw = quantize_raw(synth_stream((total + 1) * (N // 2), bool(is_real), seed=9, fft_size=N), "s16", bool(is_real))
    mk = lambda: Context(N, is_real, _levels(R), additional_size=n, audio_fft_size=n, input_format="s16",
                         max_batch=F, max_clients=2)
    a, b = mk(), mk()
    try:
        hb = a.half_frame_bytes()
        halves = raw.view(np.uint8).reshape(total + 1, hb)
        d = a.dev_alloc(raw.nbytes)
        a.h2d(d, raw)
        ga, gb = AudioClient(a), AudioClient(b)
        for g in (ga, gb):
            g.set_audio_demodulation("AM")
            g.set_audio_range(R // 3 - 100, R // 3 + 0.5, R // 3 + 100)
        b.ring_create(nh)
        pinned = [b.pinned_array(hb) for _ in range(nh)]
        with pytest.raises(PsdrError):
            b.process_ring(0, F)                    # nothing written yet

        def write(i):
            pinned[i % nh][:] = halves[i]
            b.ring_write_async(i, pinned[i % nh])

        written = 0
        for h in range(F + 1):
            write(h)
            written += 1
        for bi in range(nb):
            first = bi * F
            b.process_ring(first, F)
            b.demod_batch(first)
            # write ahead: the halves of the NEXT batch while this one is in flight (their slots were
            # read by earlier batches; the copy waits for those on the device)
            while written <= min(total, first + 2 * F):
                b.ring_wait(written - nh) if written >= nh else None   # the pinned buffer is reused
                write(written)
                written += 1
            a.process_batch(d, F, offset_bytes=first * hb)
            a.demod_batch(first)
            for f in range(F):
                assert np.array_equal(a.read_spectrum(f).view(np.uint32), b.read_spectrum(f).view(np.uint32)), (bi, f)
                assert np.array_equal(a.read_quantized(f), b.read_quantized(f)), (bi, f)
            xa, xb = ga.read_audio(), gb.read_audio()
            assert np.array_equal(xa[0].view(np.uint32), xb[0].view(np.uint32)) and np.array_equal(xa[1], xb[1])
        with pytest.raises(PsdrError):
            b.process_ring(nh - 2, F)               # would cross the ring end: must be split
        a.dev_free(d)
    finally:
        a.close()
        b.close()


def test_band_calls_refuse_bad_arguments():
    """psdr_pack_band / psdr_demod_batch_from_band: state and range errors come back as codes with a message,
    nothing is launched"""
    from phantomsdr_amd import AudioClient, Context
    N, n, F = 1 << 14, 60, 2
    ctx = Context(N, False, _levels(N), additional_size=n, audio_fft_size=n, audio_rate=12000, input_format="s16",
                  max_batch=F, max_clients=2, max_waterfall_clients=1, skip_num=1)
    try:
        L = ctx.lib
        buf = ctx.dev_alloc(F * N * 8)
        assert L.psdr_pack_band(ctx.h, 1, 0, 64, buf, 64) == -4          # before any batch
        x = synth_stream((F + 1) * (N // 2), False, seed=3, fft_size=N)
        raw = quantize_raw(x, "s16", False)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        ctx.process_batch(d, F)
        assert L.psdr_pack_band(ctx.h, F + 1, 0, 64, buf, 64) == -1      # more frames than the batch
        assert L.psdr_pack_band(ctx.h, F, N, 64, buf, 64) == -1          # first bin outside
        assert L.psdr_pack_band(ctx.h, F, 0, N + 1, buf, N + 1) == -1    # wider than the spectrum
        assert L.psdr_pack_band(ctx.h, F, 0, 64, buf, 32) == -1          # stride < bins
        assert L.psdr_pack_band(ctx.h, F, 0, 64, None, 64) == -1
        assert L.psdr_pack_band(ctx.h, F, N - 10, 64, buf, 64) == 0       # wraps at the end: fine
        c = AudioClient(ctx)
        c.set_audio_demodulation("USB")
        assert c.on_window_message(1000, 1000.0, 1030)
        assert L.psdr_demod_batch_from_band(ctx.h, buf, 64, 990, 64, F, 0) == 0        # [990, 1054) holds it
        assert L.psdr_demod_batch_from_band(ctx.h, buf, 64, 1001, 64, F, 0) == -1      # l below the band
        assert b"outside the band" in L.psdr_last_error()
        assert L.psdr_demod_batch_from_band(ctx.h, buf, 64, 980, 40, F, 0) == -1       # r beyond the band
        assert L.psdr_demod_batch_from_band(ctx.h, buf, 16, 990, 64, F, 0) == -1       # stride < bins
        assert L.psdr_demod_batch_from_band(ctx.h, buf, 64, 990, 64, F + 1, 0) == -1   # > max_batch
        ctx.synchronize()
        ctx.dev_free(buf)
        ctx.dev_free(d)
    finally:
        ctx.close()
