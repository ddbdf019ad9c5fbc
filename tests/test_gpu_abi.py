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


def test_band_layout_calls_refuse_bad_arguments():
    """psdr_set_band_layout / psdr_band_region / psdr_demod_batch_from_band_region: what the banded layout does not
    support comes back as PSDR_ERR_UNSUPPORTED (the caller falls back to psdr_pack_band), bad arguments as
    PSDR_ERR_INVALID, nothing is launched and the context keeps working"""
    from phantomsdr_amd import Context
    p, sz, u32 = C.c_void_p(), C.c_size_t(), C.c_uint32()
    small = Context(1 << 14, False, _levels(1 << 14), audio_fft_size=60, audio_rate=12000, input_format="s16", max_batch=2,
                    max_clients=2, max_waterfall_clients=1, skip_num=1)
    try:
        assert small.lib.psdr_set_band_layout(small.h, 2, 60) == -6       # natural-order spectrum: unsupported
        assert b"psdr_pack_band" in small.lib.psdr_last_error()
        assert small.lib.psdr_band_region(small.h, 0, C.byref(p), C.byref(sz), C.byref(u32), C.byref(u32)) == -4
        assert small.lib.psdr_demod_batch_from_band_region(small.h, C.c_void_p(16), 1024, 0, 1024, 1, 0) == -6
    finally:
        small.close()
    real = Context(1 << 21, True, _levels(1 << 20), audio_fft_size=360, audio_rate=12000, input_format="s16", max_batch=2,
                   max_clients=2, max_waterfall_clients=1, skip_num=1)
    try:
        assert real.lib.psdr_set_band_layout(real.h, 4, 360) == -6        # fused real layout: bands share lines
    finally:
        real.close()
    N, F = 1 << 20, 2
    ctx = Context(N, False, _levels(N), audio_fft_size=360, audio_rate=12000, input_format="s16", max_batch=F, max_clients=2,
                  max_waterfall_clients=1, skip_num=1)
    try:
        L = ctx.lib
        for bad in (0, 3, 32, -1):
            assert L.psdr_set_band_layout(ctx.h, bad, 360) == -1
        assert L.psdr_set_band_layout(ctx.h, 8, N + 1) == -1
        assert L.psdr_set_band_layout(ctx.h, 8, 360) == 0
        assert L.psdr_band_region(ctx.h, 8, C.byref(p), C.byref(sz), C.byref(u32), C.byref(u32)) == -1
        fb, nb = C.c_uint32(), C.c_uint32()
        assert L.psdr_band_region(ctx.h, 3, C.byref(p), C.byref(sz), C.byref(fb), C.byref(nb)) == 0
        assert (fb.value, nb.value, sz.value) == (3 * (N // 8), N // 8 + 1024, N // 8 + 1024) and p.value
        dp, nbins = C.c_void_p(), C.c_size_t()
        assert L.psdr_spectrum_device_ptr(ctx.h, 0, C.byref(dp), C.byref(nbins)) == -6   # a frame is not one piece any more
        reg = C.c_void_p(p.value)
        assert L.psdr_demod_batch_from_band_region(ctx.h, reg, nb.value, fb.value + 8, nb.value, F, 0) == -1   # not a whole column
        assert L.psdr_demod_batch_from_band_region(ctx.h, reg, nb.value, fb.value, nb.value - 8, F, 0) == -1
        assert L.psdr_demod_batch_from_band_region(ctx.h, reg, nb.value - 1024, fb.value, nb.value, F, 0) == -1  # stride < bins
        assert L.psdr_demod_batch_from_band_region(ctx.h, reg, nb.value, fb.value, nb.value, F + 1, 0) == -1
        assert L.psdr_demod_batch_from_band_region(ctx.h, None, nb.value, fb.value, nb.value, F, 0) == -1
        # the context still transforms and answers in k order
        x = synth_stream((F + 1) * (N // 2), False, seed=3, fft_size=N)
        raw = quantize_raw(x, "s16", False)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        ctx.process_batch(d, F)
        fo = O.FFT(N, False, _levels(N), 0, 360)
        conv = O.convert(raw, "s16").view(np.complex64).reshape(F + 1, N // 2)
        fo.load(conv[1], conv[2])
        fo.execute()
        got, want = ctx.read_spectrum(1), fo.output()
        assert np.abs(got[:N] - want[:N]).max() <= 1e-4 * np.abs(want[:N]).max()
        ctx.dev_free(d)
    finally:
        ctx.close()


def test_fetch_batch_matches_per_client_reads_and_guards_late_clients():
    """psdr_fetch_batch + psdr_fetched_audio (one synchronisation, at most four copies per batch for ALL clients: the
    granularity of src/websocket.cpp:156-185's one pass over signal_slices) hand out exactly what psdr_read_audio /
    psdr_read_pcm copy per client; a client that attached AFTER the batch was demodulated gets PSDR_ERR_NO_DATA from
    both paths instead of the previous occupant's samples (the reference would not have posted a task for it)."""
    from phantomsdr_amd import AudioClient, Context
    from phantomsdr_amd._lib import PsdrError
    N, F, n = 1 << 14, 5, 248
    ctx = Context(N, False, _levels(N), additional_size=n, audio_fft_size=n, input_format="s16", max_batch=F + 2, max_clients=4)
    lib, h = ctx.lib, ctx.h
    try:
        raw = quantize_raw(synth_stream((2 * F + 1) * (N // 2), False, seed=8, fft_size=N), "s16", False)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        ctx.set_post_chain(True)
        cl = []
        for i, mode in enumerate(["USB", "AM", "FM"]):
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            g.set_audio_range(3000 + 400 * i, 3050.0 + 400 * i, 3100 + 400 * i)
            cl.append(g)
        old = cl.pop(1)
        old_id = old.id
        ctx.process_batch(d, F)
        ctx.demod_batch(0)
        old.on_close()                       # the slot is freed after the batch ...
        late = AudioClient(ctx)              # ... and taken by a client that was not part of it
        assert late.id == old_id
        with pytest.raises(PsdrError) as e:
            late.read_audio()
        assert e.value.code == -7
        with pytest.raises(PsdrError) as e:
            late.read_pcm()
        assert e.value.code == -7
        assert lib.psdr_fetch_batch(h) == 0
        pa, pp = C.POINTER(C.c_float)(), C.POINTER(C.c_int32)()
        pw, nf = C.c_float(0), C.c_int32(0)
        assert lib.psdr_fetched_audio(h, late.id, 0, C.byref(pa), C.byref(pw), C.byref(nf), C.byref(pp)) == -7
        for g in cl:
            a, p, nan = g.read_audio()
            pcm = g.read_pcm()
            for f in range(F):
                assert lib.psdr_fetched_audio(h, g.id, f, C.byref(pa), C.byref(pw), C.byref(nf), C.byref(pp)) == 0
                assert np.array_equal(np.ctypeslib.as_array(pa, (n // 2,)).view(np.uint32), a[f].view(np.uint32))
                assert np.array_equal(np.ctypeslib.as_array(pp, (n // 2,)), pcm[f])
                assert np.float32(pw.value) == p[f] and nf.value == nan[f]
            assert lib.psdr_fetched_audio(h, g.id, F, C.byref(pa), None, None, None) == -1   # frame outside the batch
        # the next batch includes the late client: both paths serve it
        late.set_audio_range(5000, 5040.0, 5080)
        ctx.process_batch(d, F, offset_bytes=F * ctx.half_frame_bytes())
        ctx.demod_batch(F)
        a, _, _ = late.read_audio()
        assert lib.psdr_fetch_batch(h) == 0
        assert lib.psdr_fetched_audio(h, late.id, F - 1, C.byref(pa), None, None, None) == 0
        assert np.array_equal(np.ctypeslib.as_array(pa, (n // 2,)).view(np.uint32), a[F - 1].view(np.uint32))
        ctx.dev_free(d)
    finally:
        ctx.close()


def test_fetch_begin_end_pipelined_over_batches_matches_synchronous_reads():
    """The served end of the path (src/signal.cpp:283-291, src/waterfall.cpp:44-51 end in host memory): psdr_fetch_begin
    enqueues the copies of batch b behind its kernels, the caller enqueues batch b + 1, psdr_fetch_end(b) - and what
    psdr_fetched_audio / _waterfall hand out for every batch is bit for bit what a run with a full synchronisation and
    per-client reads after every batch gives (float audio, pwr, NaN flags, PCM of the post chain, waterfall rows and the
    windows they were gathered with); the pointers of batch b stay valid while batch b + 1 is in flight."""
    from phantomsdr_amd import AudioClient, Context, WaterfallClient
    N, F, n, B = 1 << 16, 6, 248, 5
    raw = quantize_raw(synth_stream((B * F + 1) * (N // 2), False, seed=21, fft_size=N), "s16", False)
    specs = [("USB", 9000, 9000.0, 9061), ("LSB", 20000, 20061.0, 20061), ("AM", 30000, 30100.5, 30200), ("FM", 41000, 41100.0, 41200)]

    def run(pipelined):
        ctx = Context(N, False, _levels(N), additional_size=n, audio_fft_size=n, input_format="s16", max_batch=F, max_clients=6,
                      max_waterfall_clients=2, skip_num=2)
        out = []
        try:
            d = ctx.dev_alloc(raw.nbytes)
            ctx.h2d(d, raw)
            ctx.set_post_chain(True)
            cl = []
            for mode, l, m, r in specs:
                g = AudioClient(ctx)
                g.set_audio_demodulation(mode)
                g.set_audio_range(l, m, r)
                cl.append(g)
            wf = [WaterfallClient(ctx), WaterfallClient(ctx)]
            wf[0].set_waterfall_range(_levels(N) - 1, 0, 1024)
            wf[1].set_waterfall_range(2, 1000, 1700)
            hb = ctx.half_frame_bytes()
            what = ctx.FETCH_AUDIO | ctx.FETCH_PCM | ctx.FETCH_WATERFALL

            def collect():
                rec = {"a": [], "w": []}
                for g in cl:
                    rec["a"].append([ctx.fetched_audio(g.id, f, pcm=True) for f in range(F)])
                for w in wf:
                    rec["w"].append(ctx.fetched_waterfall(w.id))
                return rec

            for b in range(B):
                if b == 2:  # a window change between two batches: the fetched window is the batch's, not the live one
                    wf[1].set_waterfall_range(3, 500, 900)
                    cl[0].set_audio_range(9500, 9500.0, 9561)
                ctx.process_batch(d, F, offset_bytes=b * F * hb)
                ctx.demod_batch(b * F)
                ctx.waterfall_batch(b * F)
                if pipelined:
                    ctx.fetch_begin(what)
                    if b > 0:
                        ctx.fetch_end()          # batch b - 1 lands while batch b runs
                        out.append(collect())
                else:
                    ctx.synchronize()
                    rec = {"a": [], "w": []}
                    for g in cl:
                        a, p, nan = g.read_audio()
                        pcm = g.read_pcm()
                        rec["a"].append([(a[f], p[f], nan[f], pcm[f]) for f in range(F)])
                    for w in wf:
                        rows, _ = w.read_waterfall()
                        ns, lv, l, r = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
                        assert ctx.lib.psdr_read_waterfall(ctx.h, w.id, None, 0, C.byref(ns), C.byref(lv), C.byref(l), C.byref(r)) == 0
                        rec["w"].append((rows, lv.value, l.value, r.value))
                    out.append(rec)
            if pipelined:
                ctx.fetch_end()
                out.append(collect())
                assert ctx.lib.psdr_fetch_end(ctx.h) == -4  # PSDR_ERR_STATE: nothing in flight
            ctx.dev_free(d)
        finally:
            ctx.close()
        return out

    want, got = run(False), run(True)
    assert len(want) == len(got) == B
    for b in range(B):
        for ci in range(len(specs)):
            for f in range(F):
                wa, wp, wn, wpcm = want[b]["a"][ci][f]
                ga, gp, gn, gpcm = got[b]["a"][ci][f]
                assert np.array_equal(ga.view(np.uint32), wa.view(np.uint32)), (b, ci, f)
                assert np.float32(gp) == np.float32(wp) and gn == wn
                assert np.array_equal(gpcm, wpcm), (b, ci, f)
        for wi in range(2):
            wr, wl_, wlo, whi = want[b]["w"][wi]
            gr, gl_, glo, ghi = got[b]["w"][wi]
            assert (gl_, glo, ghi) == (wl_, wlo, whi), (b, wi)
            assert np.array_equal(gr, wr), (b, wi)


def test_band_demodulation_ignores_clients_without_a_window():
    """psdr_demod_batch_from_band: a client between psdr_client_add and its first set_audio_range has the empty
    window [0, 0) - outside every band but the first - and reads no bin: it must not fail the batch for everyone
    (ADVICE r2); a real window outside the band still does."""
    import ctypes
    from phantomsdr_amd import AudioClient, Context
    N, F, n = 1 << 14, 3, 248
    ctx = Context(N, False, _levels(N), additional_size=n, audio_fft_size=n, input_format="s16", max_batch=F, max_clients=4)
    lib, h = ctx.lib, ctx.h
    try:
        raw = quantize_raw(synth_stream((F + 1) * (N // 2), False, seed=9, fft_size=N), "s16", False)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        g = AudioClient(ctx)
        g.set_audio_range(9000, 9030.0, 9060)
        ctx.process_batch(d, F)
        ctx.demod_batch(0)
        want = g.read_audio()[0].copy()
        first, bins = 8192, 4096 + n
        band = ctx.dev_alloc(F * bins * 8)
        assert lib.psdr_pack_band(h, F, first, bins, band, bins) == 0
        fresh = AudioClient(ctx)  # no window yet
        # a second context plays the receiving rank
        rx = Context(N, False, _levels(N), additional_size=n, audio_fft_size=n, input_format="s16", max_batch=F, max_clients=4)
        try:
            g2 = AudioClient(rx)
            g2.set_audio_range(9000, 9030.0, 9060)
            empty = AudioClient(rx)
            host = np.empty(F * bins * 2, np.float32)
            ctx.d2h(host, band)
            band2 = rx.dev_alloc(host.nbytes)
            rx.h2d(band2, host)
            assert rx.lib.psdr_demod_batch_from_band(rx.h, band2, bins, first, bins, F, 0) == 0, rx.lib.psdr_last_error()
            rx.last_demod_frames = F
            assert np.array_equal(g2.read_audio(F)[0].view(np.uint32), want.view(np.uint32))
            assert np.abs(empty.read_audio(F)[0]).max() == 0          # an empty slice demodulates to silence
            empty.set_audio_range(100, 130.0, 160)                    # a real window outside the band: refused
            assert rx.lib.psdr_demod_batch_from_band(rx.h, band2, bins, first, bins, F, F) != 0
            assert b"outside the band" in rx.lib.psdr_last_error()
            # ... and the refusal left the parameter ring usable
            empty.on_close()
            assert rx.lib.psdr_demod_batch_from_band(rx.h, band2, bins, first, bins, F, F) == 0
            rx.dev_free(band2)
        finally:
            rx.close()
        del fresh
        ctx.dev_free(band)
        ctx.dev_free(d)
    finally:
        ctx.close()
