"""The oracle's DFT kernels (stand-in for FFTW 3.3.10, which is not in the reference tree
and not installed here) against an independent float64 numpy DFT."""
import numpy as np
import pytest

from oracle import oracle as O


@pytest.mark.parametrize("n", [2, 4, 8, 64, 1024, 4096, 1 << 16])
def test_c2c_pow2(n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.abs(O.dft_c2c(x, -1) - ref).max() <= 4e-7 * np.abs(ref).max()
    refi = np.fft.ifft(x.astype(np.complex128)) * n
    assert np.abs(O.dft_c2c(x, +1) - refi).max() <= 4e-7 * np.abs(refi).max()


@pytest.mark.parametrize("n", [4, 12, 20, 28, 60, 124, 248, 360, 720, 1000, 3356, 10068])
def test_any_length_backward_and_c2r(n):
    """audio_fft_size is any multiple of 4 (248 = 8*31, 10068 = 4*3*839 ...)."""
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    refi = np.fft.ifft(x.astype(np.complex128)) * n
    assert np.abs(O.dft_c2c(x, +1) - refi).max() <= 2e-7 * np.abs(refi).max()
    # c2r reads bins 0..n/2 and ignores Im of bin 0 and bin n/2 (FFTW semantics)
    h = x.astype(np.complex128)[: n // 2 + 1].copy()
    h[0] = h[0].real
    h[-1] = h[-1].real
    refr = np.fft.irfft(h, n) * n
    assert np.abs(O.dft_c2r(x, n) - refr).max() <= 2e-7 * np.abs(refr).max()


@pytest.mark.parametrize("n", [16, 1024, 1 << 15])
def test_r2c(n):
    x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
    ref = np.fft.rfft(x.astype(np.float64))
    assert np.abs(O.dft_r2c(x) - ref).max() <= 4e-7 * np.abs(ref).max()


def test_linearity_and_parseval():
    n = 4096
    rng = np.random.default_rng(5)
    a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    b = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    fa, fb, fab = O.dft_c2c(a, -1), O.dft_c2c(b, -1), O.dft_c2c(a + b, -1)
    assert np.abs(fab - (fa + fb)).max() <= 1e-5 * np.abs(fab).max()
    assert abs(np.sum(np.abs(fa) ** 2) / n - np.sum(np.abs(a) ** 2)) <= 1e-5 * np.sum(np.abs(a) ** 2)
