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
