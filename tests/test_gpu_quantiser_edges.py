"""The waterfall quantiser's edges ON THE GPU (src/fft_impl.cpp:14-70): the lower clamp `std::max(-128.f, ...)` on
digital silence and on powers that straddle it, the saturation at +127 this build defines for what overflows int8 in the
reference, zero / denormal powers through the packed `v_med3` / `v_cvt_i32_f32_sdwa` path (quantize.h), and
`brightness_offset != 0` (`size_log2 = round(log2 N) + brightness_offset`, src/fft_impl.cpp:63-70) - at 2^16 points
(generic kernels), 2^17 real (three-pass real path), and the tile-major sizes 2^20 IQ, 2^21 and 2^22 real (fused
epilogues, k_col_tail, k_pyramid_tail).  int8 pyramids are bit-exact against the reference quantiser applied to the
GPU's own spectrum and within SURVEY B.3 of the oracle's."""
import numpy as np
import pytest

from helpers import quantize_raw
from oracle import oracle as O
from test_gpu_parity import levels_for

pytestmark = pytest.mark.gpu


def _level_slices(R, levels):
    off, out = 0, []
    for i in range(levels):
        out.append(slice(off, off + (R >> i)))
        off += R >> i
    return out


def _run(N, is_real, fmt, brightness, halves_float):
    """halves_float: [nh][N/2] float64 (real) or complex128 (IQ), full scale 1.0.  Returns per frame
    (q_gpu, spectrum_gpu_k_order, q_oracle, power_oracle)."""
    from phantomsdr_amd import Context
    R = N // 2 if is_real else N
    levels = levels_for(R)
    nh = len(halves_float)
    F = nh - 1
    x = np.concatenate(halves_float)
    raw = quantize_raw(x, fmt, bool(is_real))
    conv = O.convert(raw, fmt)
    halves = (conv if is_real else conv.view(np.complex64)).reshape(nh, N // 2)
    ctx = Context(N, bool(is_real), levels, brightness_offset=brightness, input_format=fmt, max_batch=F)
    out = []
    try:
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        ctx.process_batch(d, F)
        fo = O.FFT(N, bool(is_real), levels, brightness, 0)
        for f in range(F):
            fo.load(halves[f], halves[f + 1])
            fo.execute()
            out.append((ctx.read_quantized(f).copy(), ctx.read_spectrum(f).copy(), fo.quantized().copy(), fo.power().copy()))
        ctx.dev_free(d)
    finally:
        ctx.close()
    return out, levels, R


def _check_frame(qg, Xg, qo, N, is_real, levels, brightness, robust=None, tag=""):
    q_self = O.pyramid_from_spectrum(Xg, N, bool(is_real), levels, brightness)
    assert np.array_equal(qg, q_self), f"{tag}: {int((qg != q_self).sum())} of {qg.size} entries differ from the reference quantiser on the GPU's own spectrum"
    d = np.abs(qg.astype(np.int16) - qo.astype(np.int16))
    if robust is not None:
        d = d[robust]
    assert d.size == 0 or d.max() <= 1, f"{tag}: {int(d.max())} LSB from the oracle"
    assert (d != 0).sum() <= max(1, 1e-3 * d.size), f"{tag}: mismatch rate {(d != 0).mean():.2e}"


SHAPES = [(1 << 16, 0), (1 << 17, 1), (1 << 20, 0), (1 << 21, 1), (1 << 22, 1)]


@pytest.mark.parametrize("brightness", [-3, 5])
@pytest.mark.parametrize("N,is_real", SHAPES)
@pytest.mark.parametrize("fmt", ["u8", "s16"])
def test_silence_and_a_full_scale_tone(N, is_real, fmt, brightness):
    """halves: silence, silence, tone, tone, tone -> frame 0 is digital silence (u8: offset binary 128 -> exactly 0.0,
    src/samplereader.cpp:29-40): every level of the pyramid is -128, the lower clamp of src/fft_impl.cpp:40-42, 57-59.
    Frames 2, 3: an on-bin full-scale carrier - 20 log10(A^2 N / 4) + 127 is far above int8: the carrier's
    bin and its two Hann neighbours saturate at +127, at level 0 and in every level above.  Frame 1 (half silence, half
    carrier) is a broadband spectrum with both clamps in it.  Against the oracle only where its own value is not f32
    rounding noise of the transform (power >= 1e-5 of the frame's peak: an on-bin carrier leaves 140 dB of nothing)."""
    if N >= (1 << 22) and (fmt == "u8" or brightness == -3):
        pytest.skip("2^22: one format and one offset are enough")
    h = N // 2
    t = np.arange(3 * h, dtype=np.float64)
    k0 = N // 8 + 3
    fs = 127.0 / 128.0 if fmt == "u8" else 32767.0 / 32768.0
    tone = fs * (np.cos(2 * np.pi * k0 * t / N) if is_real else np.exp(2j * np.pi * k0 * t / N))
    sil = np.zeros(h, tone.dtype)
    halves = [sil, sil, tone[:h], tone[h:2 * h], tone[2 * h:]]
    res, levels, R = _run(N, is_real, fmt, brightness, halves)
    sl = _level_slices(R, levels)
    j0 = k0 if is_real else (k0 - (N // 2 + 1)) % N
    for f, (qg, Xg, qo, Po) in enumerate(res):
        tag = f"N=2^{N.bit_length() - 1} real={is_real} {fmt} brightness {brightness} frame {f}"
        if f == 0:
            assert np.all(Xg[: (N // 2 if is_real else N)] == 0)
            assert np.all(qg == -128) and np.all(qo == -128), tag
            continue
        robust = Po[: qg.size] >= 1e-5 * Po[:R].max() if Po.size >= qg.size else None
        _check_frame(qg, Xg, qo, N, is_real, levels, brightness, robust, tag)
        if f >= 2:
            for i in range(levels):
                lv = qg[sl[i]]
                assert lv[j0 >> i] == 127 and qo[sl[i]][j0 >> i] == 127, (tag, i)
            assert qg[sl[0]][j0 - 1] == 127 and qg[sl[0]][j0 + 1] == 127
            # nothing else is anywhere near - except, for 8-bit samples under a positive offset, the strongest harmonic
            # spurs of the quantised carrier itself (-60 dBc is 120 LSB below a peak that sits 140 LSB above the scale)
            assert 3 <= (qg[sl[0]] == 127).sum() <= (3 if fmt == "s16" else 24), (tag, int((qg[sl[0]] == 127).sum()))


@pytest.mark.parametrize("brightness", [-3, 5])
@pytest.mark.parametrize("N,is_real", SHAPES[:4])
def test_powers_around_the_lower_clamp_and_in_the_denormal_range(N, is_real, brightness):
    """f32 input.  Frame 0: white noise at 2^-20 FS - 20 log10(0.75 sigma^2 2^b) + 127 is about -115 + 6 b: the lower
    clamp cuts through the distribution (b = -3) or sits just below it (b = +5).  Frame 2: the same noise at 2^-63 FS:
    powers around 2^-126 / N, the denormal range of f32 and below it (zero) - everything is -128 whatever a
    flush-to-zero mode does to the products.  Frame 1 mixes the two halves."""
    h = N // 2
    rng = np.random.default_rng(11)
    nz = rng.standard_normal(4 * h) if is_real else (rng.standard_normal(4 * h) + 1j * rng.standard_normal(4 * h))
    a, b = 2.0 ** -20, 2.0 ** -63
    halves = [a * nz[:h], a * nz[h:2 * h], b * nz[2 * h:3 * h], b * nz[3 * h:]]
    res, levels, R = _run(N, is_real, "f32", brightness, halves)
    for f, (qg, Xg, qo, Po) in enumerate(res):
        tag = f"N=2^{N.bit_length() - 1} real={is_real} f32 brightness {brightness} frame {f}"
        _check_frame(qg, Xg, qo, N, is_real, levels, brightness, None, tag)
        lv0 = qg[:R]
        if f == 0:
            if brightness == -3:
                assert 0.001 < (lv0 == -128).mean() < 0.999, (tag, (lv0 == -128).mean())   # the clamp cuts through
            else:
                assert (lv0 > -128).mean() > 0.9
        if f == 2:
            assert np.all(qg == -128) and np.all(qo == -128), tag


def test_brightness_offset_shifts_every_level_by_six_lsb_per_step():
    """size_log2 enters the quantiser as an exponent offset (src/fft_impl.cpp:14-23, 63-70): one step of
    brightness_offset is 20 log10(2) = 6.02 LSB on every unclamped entry of every level."""
    N, h = 1 << 16, 1 << 15
    rng = np.random.default_rng(3)
    nz = (rng.standard_normal(2 * h) + 1j * rng.standard_normal(2 * h)) * 2.0 ** -9
    q = {}
    for b in (0, 4):
        res, levels, R = _run(N, 0, "s16", b, [nz[:h], nz[h:]])
        q[b] = res[0][0].astype(np.int32)
    ok = (q[0] > 0) & (q[4] < 127)   # (positive values: truncation toward zero is floor on both sides)
    d = (q[4] - q[0])[ok]
    assert ok.mean() > 0.8 and d.min() >= 24 and d.max() <= 25 and abs(d.mean() - 4 * 6.0206) < 0.05
