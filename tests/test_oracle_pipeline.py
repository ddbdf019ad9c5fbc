"""An independent numpy restatement (float64 DFTs, SURVEY Appendix B formulas) of the whole
path against the C oracle: framing + window, spectrum + /N, rotation map, int8 pyramid,
per-mode demodulation with overlap-add, DC blocker."""
import numpy as np
import pytest

from helpers import rel_err, rel_l2, synth_stream
from oracle import oracle as O

f32 = np.float32


def fma32(a, b, c):
    """float32 fused multiply-add via float64 (the 24x24-bit product is exact in double)"""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def np_quantize(P, off):
    """src/fft_impl.cpp:14-23,40-42 with the FMA placement of the reference's flags"""
    P = np.ascontiguousarray(P, f32)
    bits = P.view(np.uint32)
    e = ((bits >> 23) & 0xFF).astype(np.int32) - 128
    m = ((bits & np.uint32(0x007FFFFF)) | np.uint32(0x3F800000)).view(f32)
    logv = e.astype(f32) + f32(off)
    t = fma32(np.full_like(m, f32(-0.34484843)), m, np.full_like(m, f32(2.02466578)))
    poly = fma32(t, m, np.full_like(m, f32(-0.67487759)))
    logv = (logv + poly).astype(f32)
    v = (logv * f32(0.3010299956639812)).astype(f32)
    q = fma32(v, np.full_like(v, f32(20.0)), np.full_like(v, f32(127.0)))
    c = np.where(q > -128.0, q, f32(-128.0))
    return np.where(c >= 127.0, 127, np.trunc(c)).astype(np.int8)


@pytest.mark.parametrize("N,is_real", [(4096, 0), (8192, 1), (1 << 14, 0), (1 << 15, 1)])
def test_forward_and_pyramid(N, is_real):
    levels = 3
    x = synth_stream(N, is_real, seed=N + is_real, fft_size=N)
    x = x.astype(f32) if is_real else x.astype(np.complex64)
    fo = O.FFT(N, is_real, levels, 0, 64)
    fo.load(x[: N // 2], x[N // 2:])
    fo.execute()
    # B.1/B.2: windowed frame, DFT, /N
    w = O.hann(N)
    xin = (x * w).astype(x.dtype)
    X = (np.fft.rfft(xin.astype(np.float64)) if is_real else np.fft.fft(xin.astype(np.complex128)))
    out = fo.output()
    if is_real:
        assert rel_err(out[: N // 2] * N, X[: N // 2]) < 1e-6
        assert abs(out[N // 2] - X[N // 2]) <= 1e-6 * np.abs(X).max()       # never normalised
        spec_c = out[: N // 2]
    else:
        assert rel_err(out[:N] * N, X) < 1e-6
        assert np.array_equal(out[N:N + 64], out[:64])                          # wrap copy
        base = N // 2 + 1
        spec_c = out[(np.arange(N) + base) % N]                                 # client order
    # B.3 from the oracle's own f32 spectrum: bit-exact up to rare double roundings in fma32
    R = N // 2 if is_real else N
    P = fma32(spec_c.real.astype(f32), spec_c.real.astype(f32), (spec_c.imag.astype(f32) ** 2).astype(f32))
    q = fo.quantized()
    off = 0
    s = int(round(np.log2(N)))
    for lv in range(levels):
        qq = np_quantize(P, s - lv)
        got = q[off: off + (R >> lv)]
        assert (qq != got).mean() <= 1e-4 and np.abs(qq.astype(int) - got.astype(int)).max() <= 1
        off += R >> lv
        P = (P[0::2] + P[1::2]).astype(f32)


def np_client_frame(state, S, l, mid, r, mode, n, frame, is_real):
    """B.4 in float64.  state: dict(real_prev, B).  S = spectrum slice [l, r)."""
    m = int(np.floor(mid)) - l
    m_idx = int(np.floor(mid))
    ln = r - l
    A = np.zeros(n, np.complex128)
    odd = (m_idx % 2 == 1) if m_idx >= 0 else False   # C++ remainder semantics
    flip = (frame % 2 == 1) and (((m_idx % 2 == 0) and not is_real) or (odd and is_real))
    if mode in ("USB", "LSB"):
        if mode == "USB":
            for t in range(max(0, m), min(ln, m + n)):
                A[t - m] = S[t]
        else:
            for t in range(max(0, m - n + 1), min(ln, m + 1)):
                A[m - t] = S[t]
        h = A[: n // 2 + 1].copy()
        h[0] = h[0].real
        h[-1] = h[-1].real
        y = np.fft.irfft(h, n) * n
        if mode == "LSB":
            y = y[::-1]
        if flip:
            y = -y
        out = y[: n // 2] + state["real_prev"]
        state["real_prev"] = y[n // 2:].copy()
        return out
    for t in range(max(0, m), min(ln, m + n // 2)):
        A[t - m] = S[t]
    for t in range(max(0, m - n // 2 + 1), min(ln, m)):
        A[n - m + t] = S[t]
    last = state["B"][n // 2 - 1]
    Bprev = state["B"][n // 2:].copy()
    B = np.fft.ifft(A) * n
    if flip:
        B = -B
    B[: n // 2] += Bprev
    state["B"] = B
    if mode == "AM":
        return np.abs(B[: n // 2])
    prev = np.concatenate([[last], B[: n // 2 - 1]])
    return np.angle(B[: n // 2] * np.conj(prev))


@pytest.mark.parametrize("N,is_real,n", [(4096, 0, 60), (8192, 1, 60), (1 << 14, 0, 248)])
def test_clients_all_modes(N, is_real, n):
    R = N // 2 if is_real else N
    nframes = 7
    x = synth_stream((nframes + 1) * (N // 2), is_real, seed=77 + is_real, fft_size=N)
    x = (x.astype(f32) if is_real else x.astype(np.complex64)).reshape(nframes + 1, N // 2)
    am = int((0.11 * N) if is_real else ((0.11 * N - (N // 2 + 1)) % N))
    w, h = n // 4, n // 2 - 2
    specs = [("USB", am, float(am), am + w), ("USB", am + 1, am + 1.5, am + 1 + w),
             ("LSB", am - w, float(am), am), ("LSB", am - w + 1, am + 1.25, am + 1),
             ("AM", am - h, float(am), am + h), ("FM", am - h + 1, am + 1.0, am + h),
             ("USB", 0, 0.0, w), ("LSB", R - 1 - w, float(R - 1), R - 1), ("AM", 40, 300.0, 60)]
    if not is_real:
        dc = N // 2 - 1
        specs.append(("USB", dc - 10, dc - 10.0, dc + w))          # crosses the k-space wrap
    fo = O.FFT(N, is_real, 3, 0, n)
    ocl, st = [], []
    for mode, l, m, r in specs:
        c = O.AudioClient(is_real, n, 12000, R)
        c.set_audio_demodulation(mode)
        c.set_audio_range(l, m, r)
        ocl.append(c)
        st.append({"real_prev": np.zeros(n // 2), "B": np.zeros(n, np.complex128)})
    for f in range(nframes):
        fo.load(x[f], x[f + 1])
        fo.execute()
        out = fo.output().copy()
        spec_c = out[: N // 2] if is_real else out[(np.arange(N) + N // 2 + 1) % N]
        for ci, (mode, l, m, r) in enumerate(specs):
            a_o, p_o, _, dropped = ocl[ci].send_audio(out, f, fft=fo)
            assert not dropped
            S = spec_c[l:r].astype(np.complex128)
            a_n = np_client_frame(st[ci], S, l, m, r, mode, n, f, is_real)
            assert abs(p_o - np.sum(np.abs(S) ** 2)) <= 1e-5 * max(np.sum(np.abs(S) ** 2), 1e-30)
            scale = max(np.abs(a_n).max(), 1e-30)
            if mode == "FM":
                assert np.abs(np.angle(np.exp(1j * (a_o - a_n)))).max() < 1e-3
            else:
                assert np.abs(a_o - a_n).max() <= 2e-5 * scale, (mode, f)


def test_dc_blocker_against_definition():
    """DCBlocker (src/utils.h:139-169): delayed sample minus the mean of the means, f32 sums."""
    delay = 32
    rng = np.random.default_rng(9)
    x = (rng.standard_normal(2000) * 0.1 + 0.3).astype(f32)
    d = O.lib().orc_dc_create(delay)
    y = x.copy()
    O.lib().orc_dc_remove(d, O._p(y), y.size)
    q1, q2 = np.zeros(delay, f32), np.zeros(delay, f32)
    s1 = s2 = f32(0)
    exp = np.zeros_like(x)
    for i, v in enumerate(x):
        s1 = f32(s1 + (-q1[-1]))
        q1 = np.concatenate([[v], q1[:-1]])
        s1 = f32(s1 + v)
        ma1 = f32(s1 / f32(delay))
        s2 = f32(s2 + (-q2[-1]))
        q2 = np.concatenate([[ma1], q2[:-1]])
        s2 = f32(s2 + ma1)
        exp[i] = f32(q1[delay - 1] - f32(s2 / f32(delay)))
    assert np.array_equal(y, exp)
    assert abs(np.mean(y[500:])) < 5e-3   # the DC offset is gone


def test_convert_formats():
    """convert<T,T_signed> (src/samplereader.cpp:29-40): unsigned flips the MSB, then /2^(bits-1)."""
    u8 = np.arange(256, dtype=np.uint8)
    assert np.array_equal(O.convert(u8, "u8"), ((u8.astype(np.int16) - 128) / 128.0).astype(f32))
    s8 = np.arange(-128, 128, dtype=np.int8)
    assert np.array_equal(O.convert(s8, "s8"), (s8 / 128.0).astype(f32))
    u16 = np.array([0, 1, 32767, 32768, 65535], np.uint16)
    assert np.array_equal(O.convert(u16, "u16"), ((u16.astype(np.int32) - 32768) / 32768.0).astype(f32))
    s16 = np.array([-32768, -1, 0, 1, 32767], np.int16)
    assert np.array_equal(O.convert(s16, "s16"), (s16 / 32768.0).astype(f32))
    f = np.array([0.5, -1.25, 3e-9], np.float64)
    assert np.array_equal(O.convert(f, "f64"), f.astype(f32))
    assert np.array_equal(O.convert(f.astype(f32), "f32"), f.astype(f32))


def test_quantiser_edges():
    """lower clamp only in the reference (src/fft_impl.cpp:40-42); the build saturates at +127."""
    assert O.quantize(np.array([0.0], f32), 20)[0] == -128
    assert O.quantize(np.array([1e30], f32), 20)[0] == 127
    # vec_log2 works on the bit pattern: inf/NaN (exponent 255) come out as a huge finite value
    assert O.quantize(np.array([np.nan, np.inf], f32), 20).tolist() == [127, 127]
    # 0.5 dB of power per LSB: doubling the power adds ~6 LSB
    a = O.quantize(np.array([1e-9, 2e-9], f32), 20)
    assert 5 <= int(a[1]) - int(a[0]) <= 7


def test_waterfall_level_choice():
    """WaterfallClient::on_window_message (src/waterfall.cpp:53-94)."""
    levels, mwf = 11, 1024
    assert O.waterfall_pick_level(levels, mwf, 0, 1 << 20) == (10, 0, 1024)
    lv, l, r = O.waterfall_pick_level(levels, mwf, 1000, 1000 + 2048)
    assert lv == 1 and (l, r) == (500, 1524)
    assert O.waterfall_pick_level(levels, mwf, 5, 5)[0] == -1
    assert O.waterfall_pick_level(levels, mwf, -1, 5)[0] == -1
