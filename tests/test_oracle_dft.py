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


@pytest.mark.parametrize("N,is_real", [(1 << 14, 0), (1 << 15, 1), (1 << 20, 0), (1 << 21, 1)])
def test_builtin_transform_against_an_fftw3_api_library(N, is_real):
    """the oracle's own forward transform against an independent production FFT reached through the
    FFTW3 API the reference calls (src/fft_impl.cpp:89-117,145): libfftw3f if installed, else MKL's
    wrappers.  Same window, same normalisation, same pyramid code - only the transform differs."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(N % 1000 + is_real)
    if is_real:
        h = (rng.standard_normal((2, N // 2)) * 1e-2).astype(np.float32)
    else:
        h = ((rng.standard_normal((2, N // 2)) + 1j * rng.standard_normal((2, N // 2))) * 1e-2).astype(np.complex64)
    O.use_fft_library("")
    f0 = O.FFT(N, is_real, 5, 0, 8)
    f0.load(h[0], h[1])
    f0.execute()
    X0, q0 = f0.output().copy(), f0.quantized().copy()
    name = O.use_fft_library()
    try:
        if not name:
            pytest.skip("no library with the FFTW3 API on this host")
        f1 = O.FFT(N, is_real, 5, 0, 8)
        f1.load(h[0], h[1])
        f1.execute()
        X1, q1 = f1.output().copy(), f1.quantized().copy()
    finally:
        O.use_fft_library("")
    nb = N // 2 if is_real else N
    assert np.abs(X0[:nb] - X1[:nb]).max() <= 2e-6 * np.abs(X1[:nb]).max()
    assert np.linalg.norm(X0[:nb] - X1[:nb]) <= 1e-6 * np.linalg.norm(X1[:nb])
    d = np.abs(q0.astype(np.int16) - q1.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3
