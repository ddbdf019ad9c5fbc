"""Pins the CORE of the oracle - quantiser + pyramid + rotation (src/fft_impl.cpp:14-61, 144-174), sample conversion
(src/samplereader.cpp:29-70), the DC blocker (src/utils.h:76-169) - against the reference's own compiled code, when the
image can build it: `make -C oracle ref_core` compiles fft_impl.cpp / samplereader.cpp / utils.h in place ONLY with a
genuine fftw3.h + libfftw3f + boost (nothing is shimmed; VERDICT r4 next #4b).  In an image without them every test
here SKIPS, and those parts of the oracle stay "parity unpinned" (oracle/psdr_oracle.h)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from helpers import quantize_raw, synth_stream
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RC = O.ref_core()
needs_ref_core = pytest.mark.skipif(RC is None, reason="oracle/_ref/libref_core.so not built: this image has no genuine fftw3.h / libfftw3f / boost "
                                                       "(make -C oracle ref_core says which)")
p = O._p


@needs_ref_core
@pytest.mark.parametrize("N,is_real", [(1 << 12, 0), (1 << 13, 1), (1 << 16, 0), (1 << 17, 1), (1 << 20, 0), (1 << 21, 1)])
def test_fft_execute_quantiser_pyramid_and_rotation_bit_exact(N, is_real):
    """FFTW::execute on the reference's side, the oracle's quantiser / pyramid / rotation applied to the REFERENCE'S OWN
    spectrum: every int8 of every level identical (the transform itself - FFTW's summation order - is not under test: the
    oracle's spectrum is compared within the f32 bound)"""
    R = N // 2 if is_real else N
    levels, n, bo = 0, 248, 0
    while (R >> levels) >= 1024:
        levels += 1
    levels = max(levels, 1)
    x = synth_stream(3 * (N // 2), bool(is_real), seed=N + is_real, fft_size=N)
    x = (x.astype(np.float32) if is_real else x.astype(np.complex64)).reshape(3, N // 2)
    h = RC.refc_fft_create(N, is_real, 1, levels, bo, 0 if is_real else n)
    fo = O.FFT(N, bool(is_real), levels, bo, n)
    qlen = sum(R >> i for i in range(levels))
    try:
        for f in range(2):
            a1, a2 = O.aligned(x[f].size * (1 if is_real else 2), np.float32), O.aligned(x[f].size * (1 if is_real else 2), np.float32)
            a1[:] = x[f].view(np.float32)
            a2[:] = x[f + 1].view(np.float32)
            assert RC.refc_fft_execute(h, is_real, p(a1), p(a2)) == 0
            nout = (N // 2 + 1) if is_real else (N + n)
            spec_ref = np.ctypeslib.as_array(C.cast(RC.refc_fft_output(h), C.POINTER(C.c_float)), shape=(2 * nout,)).view(np.complex64).copy()
            q_ref = np.ctypeslib.as_array(C.cast(RC.refc_fft_quantized(h), C.POINTER(C.c_int8)), shape=(qlen,)).copy()
            assert np.array_equal(O.pyramid_from_spectrum(spec_ref, N, bool(is_real), levels, bo), q_ref)
            if not is_real:
                assert np.array_equal(spec_ref[N:N + n], spec_ref[:n])  # the wrap copy (src/fft.cpp:91-98 does it; FFTW::execute leaves room)
            fo.load(x[f], x[f + 1])
            fo.execute()
            nb = N // 2 if is_real else N
            so = fo.output()[:nb]
            assert np.abs(so - spec_ref[:nb]).max() <= 2e-6 * np.abs(spec_ref[:nb]).max()
            d = np.abs(fo.quantized().astype(np.int16) - q_ref.astype(np.int16))
            assert d.max() <= 1 and (d != 0).mean() < 1e-3
    finally:
        RC.refc_fft_destroy(h)


@needs_ref_core
@pytest.mark.parametrize("fmt", ["u8", "s8", "u16", "s16", "f32", "f64"])
def test_sample_conversion_bit_exact(fmt):
    rng = np.random.default_rng(11)
    num = 1 << 14
    raw = quantize_raw(rng.uniform(-1, 1, num), fmt, True)
    if fmt in ("u8", "u16", "s8", "s16"):  # both ends of the range
        info = np.iinfo(raw.dtype)
        raw[:2] = [info.min, info.max]
    out = O.aligned(num, np.float32)
    assert RC.refc_convert(O.FMT[fmt], p(np.ascontiguousarray(raw)), p(out), num) == 0
    assert np.array_equal(O.convert(raw, fmt).view(np.uint32), np.array(out).view(np.uint32))


@needs_ref_core
def test_dc_blocker_bit_exact():
    rng = np.random.default_rng(12)
    L = O.lib()
    rd, od = RC.refc_dc_create(32), L.orc_dc_create(32)
    try:
        for it in range(40):
            s = (rng.standard_normal(180) * 0.1 + 0.3).astype(np.float32)
            a, b = O.aligned(180, np.float32), s.copy()
            a[:] = s
            RC.refc_dc_process(rd, p(a), 180)
            L.orc_dc_remove(od, p(b), 180)
            assert np.array_equal(np.array(a).view(np.uint32), b.view(np.uint32)), it
    finally:
        RC.refc_dc_destroy(rd)
        L.orc_dc_destroy(od)


def test_ref_core_doorways_compile_against_the_reference_headers():
    """the doorway file is plain C++ over the reference's own declarations: a -fsyntax-only pass against the real
    src/fft.h and src/samplereader.h (with a one-line <fftw3.h> in a temporary directory for THIS CHECK ONLY - the build
    target never sees it, and without a genuine fftw3.h it builds nothing)"""
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("no reference tree")
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "fftw3.h"), "w") as f:
            f.write("typedef struct fftwf_plan_s *fftwf_plan; typedef float fftwf_complex[2]; enum { FFTW_ESTIMATE = 1 << 6 };\n")
        r = subprocess.run(["g++", "-std=c++20", "-fsyntax-only", "-w", "-I" + d, "-I/root/reference/src",
                            os.path.join(ROOT, "oracle", "ref_core_exports.cpp")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]


def test_ref_core_target_never_builds_from_stand_ins():
    """the probe asks the compiler for the genuine headers and libraries; in an image without them the target prints why
    and leaves no library behind"""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref_core"], capture_output=True, text=True)
    assert r.returncode == 0
    built = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_core.so"))
    assert built == ("built _ref/libref_core.so" in r.stdout) or (built and RC is not None)
    if not built:
        assert "ref_core:" in r.stdout and ("nothing is shimmed" in r.stdout or "absent" in r.stdout)
