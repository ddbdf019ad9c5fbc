"""Shared synthetic-signal generators and comparison helpers for the tests."""
import numpy as np

FMT_DTYPE = {"u8": np.uint8, "s8": np.int8, "u16": np.uint16, "s16": np.int16,
             "f32": np.float32, "f64": np.float64}


def synth_stream(nsamples, is_real, seed, sigma=2.0 ** -9, ntones=8, fft_size=None):
    """white Gaussian noise + a few CW tones + one AM and one FM carrier, float64.
    returns real[nsamples] or complex[nsamples]."""
    rng = np.random.default_rng(seed)
    N = fft_size or nsamples
    t = np.arange(nsamples, dtype=np.float64)
    if is_real:
        x = rng.standard_normal(nsamples) * sigma
        amp = 2.0 / np.sqrt(N)
        for _ in range(ntones):
            f = rng.uniform(0.02, 0.48)
            x += amp * rng.uniform(0.3, 1.0) * np.cos(2 * np.pi * f * t + rng.uniform(0, 6.28))
        fa = 0.11
        x += amp * (1 + 0.5 * np.cos(2 * np.pi * 3e-4 * t)) * np.cos(2 * np.pi * fa * t)
        ff = 0.31
        x += amp * np.cos(2 * np.pi * ff * t + 5.0 * np.sin(2 * np.pi * 2e-4 * t))
    else:
        x = (rng.standard_normal(nsamples) + 1j * rng.standard_normal(nsamples)) * sigma
        amp = 1.0 / np.sqrt(N)
        for _ in range(ntones):
            f = rng.uniform(-0.48, 0.48)
            x += amp * rng.uniform(0.3, 1.0) * np.exp(1j * (2 * np.pi * f * t + rng.uniform(0, 6.28)))
        x += amp * (1 + 0.5 * np.cos(2 * np.pi * 3e-4 * t)) * np.exp(2j * np.pi * 0.11 * t)
        x += amp * np.exp(1j * (2 * np.pi * -0.21 * t + 5.0 * np.sin(2 * np.pi * 2e-4 * t)))
    return x


def quantize_raw(x, fmt, is_real):
    """float64 stream (full scale = 1.0) -> raw samples of `fmt` (interleaved I/Q for IQ)."""
    if not is_real:
        x = np.stack([x.real, x.imag], axis=-1).reshape(-1)
    if fmt == "u8":
        return (np.clip(np.round(x * 128), -128, 127).astype(np.int16) + 128).astype(np.uint8)
    if fmt == "s8":
        return np.clip(np.round(x * 128), -128, 127).astype(np.int8)
    if fmt == "u16":
        return (np.clip(np.round(x * 32768), -32768, 32767).astype(np.int32) + 32768).astype(np.uint16)
    if fmt == "s16":
        return np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)
    if fmt == "f32":
        return x.astype(np.float32)
    if fmt == "f64":
        return x.astype(np.float64)
    raise ValueError(fmt)


def rel_err(a, b):
    """max |a-b| / max |b|"""
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


FM_TOL_RAD = 1e-4


FM_FWD_EPS = 1e-6  # forward-transform error per bin, relative to the rms of the whole spectrum (the spectrum tests grant 1e-5)


def fm_tolerance(baseband, prev, floor=2.5e-2, fwd_scale=0.0):
    """Per-sample bound for the FM discriminator output arg(B[i] * conj(B[i-1]))
    (src/utils/dsp.cpp:27-35).  A baseband error dB turns into an angle error of about
    |dB| / |B[i]| + |dB| / |B[i-1]|: BOTH samples carry their own error.  With ONE absolute baseband
    error budget E = 1e-4 * floor * max|B| = 2.5e-6 * max|B| for every sample (40 times tighter than
    the 1e-4 relative L2 that B.6 grants the other modes' audio) the bound is
        max(1e-4, E / |B[i]| + E / |B[i-1]|):
    SURVEY B.6's 1e-4 rad wherever both samples are at least 2 * floor = 5 % of the peak (or one is
    strong and the other at least 2.5 %), growing with 1 / |B| below that.  Measured on MI355X: two
    f32 evaluations of a 720-point inverse transform on two f32 forward transforms differ by up to
    1.2e-6 * max|B| (test_demod_fixed_plans_all_modes[720-1]), which is why the floor is not 1e-2.
    (Round 3's first form divided by min(|B[i]|, |B[i-1]|) only - half the budget its own derivation
    grants when both samples are weak; tools/fuzz_parity.py found the case after 147 clean ones: a
    60-point transform, both samples at 2.6 % of the peak, 1.12e-4 rad.)
    fwd_scale (oracle.AudioClient.fwd_scale: rms of the WHOLE spectrum x sqrt(bins in the window)): the budget
    above is the inverse transform's; the forward transform's own f32 rounding is proportional to what flows
    through its butterflies - the whole frame, strong carriers included - not to this window's level, so a window
    that holds only noise 20-50 dB below the carriers inherits a relative error that no implementation in f32 can
    avoid (the reference's FFTW included).  FM_FWD_EPS * fwd_scale is added to E: ten times tighter than the 1e-5
    relative L2 the spectrum tests grant, without effect (< 10 % of E) unless the window is far weaker than the
    frame's rms (found by tools/fuzz_parity.py: a window of 10 noise bins at the Nyquist edge of a frame with eight
    carriers, 1.48 x the bound without it).
    baseband: the oracle's B[0..n/2) of this frame; prev: B[n/2-1] of the previous frame."""
    mag = np.abs(np.asarray(baseband, np.complex128))
    pm = np.concatenate([[abs(complex(prev))], mag[:-1]])
    peak = max(float(mag.max()), float(pm.max()), 1e-300)
    budget = FM_TOL_RAD * floor * peak + FM_FWD_EPS * float(fwd_scale)
    return np.maximum(FM_TOL_RAD, budget / np.maximum(mag, 1e-300) + budget / np.maximum(pm, 1e-300))


def pwr_tolerance(p_ref, fwd_scale=0.0):
    """bound for a window's power sum (src/signal.cpp:117-119): 1e-4 relative, plus what the forward transform's own
    rounding (FM_FWD_EPS of the whole frame's rms per bin, see fm_tolerance) does to sqrt(p): a window that is one
    nearly empty bin - the DC bin of a real-input frame, found by tools/fuzz_parity.py - has no digits of its own"""
    e = FM_FWD_EPS * float(fwd_scale)
    return 1e-4 * max(abs(float(p_ref)), 1e-30) + 2.0 * np.sqrt(max(float(p_ref), 0.0)) * e + e * e


def fm_angle_error(a_gpu, a_ref):
    """|a_gpu - a_ref| on the circle"""
    return np.abs(np.angle(np.exp(1j * (np.asarray(a_gpu, np.float64) - np.asarray(a_ref, np.float64)))))


def check_fm(a_gpu, a_ref, baseband, prev, tag="", fwd_scale=0.0):
    dd = fm_angle_error(a_gpu, a_ref)
    tol = fm_tolerance(baseband, prev, fwd_scale=fwd_scale)
    bad = dd > tol
    assert not bad.any(), (f"{tag}: FM error {dd[bad].max():.2e} rad above the conditioned bound at {int(bad.sum())} "
                           f"samples (worst ratio {float((dd / tol).max()):.2f})")
