"""Pins the oracle against the REFERENCE'S OWN compiled code (oracle/_ref, built in place
from /root/reference/src/utils/{dsp,audioprocessing}.cpp with the reference's flags):
bit-exact for the Hann window, AM envelope, FM discriminator, float->int16 and the AGC."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

R = O.ref()
pytestmark = pytest.mark.skipif(R is None, reason="oracle/_ref was not built (no reference tree)")
p = O._p


@pytest.mark.parametrize("n", [16, 1000, 1 << 16, 1 << 20])
def test_hann_window(n):
    a = O.aligned(n, np.float32)
    R.ref_build_hann_window(p(a), n)
    assert np.array_equal(a, O.hann(n))
    assert a[0] == 0.0 and abs(a[n // 2] - 1.0) < 1e-6


def test_am_fm_int16_bit_exact():
    rng = np.random.default_rng(3)
    n = 50000
    z = O.aligned(2 * n, np.float32)
    z[:] = rng.standard_normal(2 * n).astype(np.float32) * 0.1
    o1, o2 = O.aligned(n, np.float32), np.zeros(n, np.float32)
    L = O.lib()
    R.ref_dsp_am_demod(p(z), p(o1), n)
    L.orc_am_demod(p(z), p(o2), n)
    assert np.array_equal(o1, o2)
    R.ref_polar_discriminator_fm(p(z), 0.3, -0.2, p(o1), n)
    L.orc_polar_discriminator_fm(p(z), 0.3, -0.2, p(o2), n)
    assert np.array_equal(o1, o2)
    x = O.aligned(n, np.float32)
    x[:] = rng.standard_normal(n).astype(np.float32) * 3
    i1, i2 = O.aligned(n, np.int32), np.zeros(n, np.int32)
    R.ref_dsp_float_to_int16(p(x), p(i1), 16384.0, n)
    L.orc_float_to_int16(p(x), p(i2), 16384.0, n)
    assert np.array_equal(i1, i2)
    assert i1.max() == 32767 and i1.min() == -32768  # clamps exercised


def test_negate_add_helpers():
    rng = np.random.default_rng(4)
    n = 4096
    a = O.aligned(n, np.float32)
    b = O.aligned(n, np.float32)
    a[:] = rng.standard_normal(n)
    b[:] = rng.standard_normal(n)
    a0 = a.copy()
    R.ref_dsp_negate_float(p(a), n)
    assert np.array_equal(a, -a0)
    R.ref_dsp_add_float(p(a), p(b), n)
    assert np.array_equal(a, -a0 + b)
    R.ref_dsp_negate_complex(p(a), n // 2)
    R.ref_dsp_add_complex(p(a), p(b), n // 2)
    assert np.array_equal(a, -(-a0 + b) + b)


def test_agc_bit_exact_including_reset():
    rng = np.random.default_rng(7)
    L = O.lib()
    ra = R.ref_agc_create(0.2, 50.0, 300.0, 200.0, 12000.0)
    oa = L.orc_agc_create(0.2, 50.0, 300.0, 200.0, 12000.0)
    for it in range(120):
        s = (rng.standard_normal(180) * (0.01 + 0.5 * (it % 7 == 0))).astype(np.float32)
        s1 = O.aligned(180, np.float32)
        s1[:] = s
        s2 = s.copy()
        R.ref_agc_process(ra, p(s1), 180)
        L.orc_agc_process(oa, p(s2), 180)
        assert np.array_equal(s1, s2), it
        if it == 60:
            R.ref_agc_reset(ra)
            L.orc_agc_reset(oa)
    R.ref_agc_destroy(ra)
    L.orc_agc_destroy(oa)
