"""Golden fixtures (tests/golden/*.npz, written by tools/gen_golden.py): the oracle must
keep reproducing them bit for bit, and (on the GPU) the HIP path must match them within
the parity tolerances.  The fixtures are data only and travel to the GPU box."""
import glob
import os

import numpy as np
import pytest

from helpers import check_fm, rel_err, rel_l2
from oracle import oracle as O

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load(path):
    d = np.load(path, allow_pickle=False)
    g = {k: d[k] for k in d.files}
    g["fmt"] = str(g["fmt"])
    for k in ("N", "is_real", "n", "levels"):
        g[k] = int(g[k])
    g["clients"] = [(str(m), int(l), float(mm), int(r)) for m, l, mm, r in
                    zip(g["client_modes"], g["client_l"], g["client_m"], g["client_r"])]
    return g


def test_fixtures_present():
    assert len(FIXTURES) >= 4


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_fixture(path):
    g = load(path)
    N, is_real, n, levels = g["N"], g["is_real"], g["n"], g["levels"]
    R = N // 2 if is_real else N
    conv = O.convert(g["raw"], g["fmt"])
    nframes = g["spectrum"].shape[0]
    halves = (conv if is_real else conv.view(np.complex64)).reshape(nframes + 1, N // 2)
    fo = O.FFT(N, is_real, levels, 0, n)
    ocl = []
    for mode, l, m, r in g["clients"]:
        c = O.AudioClient(is_real, n, 12000, R)
        c.set_audio_demodulation(mode)
        c.set_audio_range(l, m, r)
        ocl.append(c)
    for f in range(nframes):
        fo.load(halves[f], halves[f + 1])
        fo.execute()
        assert np.array_equal(fo.output().view(np.uint32), g["spectrum"][f].view(np.uint32))
        assert np.array_equal(fo.quantized(), g["quantized"][f])
        for ci, c in enumerate(ocl):
            a, p, pcm, _ = c.send_audio(fo.output(), f, fft=fo, post=True)
            assert np.array_equal(a.view(np.uint32), g["audio"][ci, f].view(np.uint32))
            assert np.float32(p) == g["pwr"][ci, f]
            assert np.array_equal(pcm, g["pcm"][ci, f])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_matches_fixture(path):
    from phantomsdr_amd import AudioClient, Context
    g = load(path)
    N, is_real, n, levels = g["N"], g["is_real"], g["n"], g["levels"]
    nframes = g["spectrum"].shape[0]
    nb = N // 2 if is_real else N
    ctx = Context(N, is_real, levels, additional_size=n, audio_fft_size=n, input_format=g["fmt"],
                  max_batch=nframes, max_clients=len(g["clients"]))
    try:
        raw = g["raw"]
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        cls = []
        for mode, l, m, r in g["clients"]:
            c = AudioClient(ctx)
            c.set_audio_demodulation(mode)
            c.set_audio_range(l, m, r)
            cls.append(c)
        ctx.process_batch(d, nframes)
        ctx.demod_batch(0)
        for f in range(nframes):
            Xg = ctx.read_spectrum(f)
            assert rel_err(Xg[:nb], g["spectrum"][f][:nb]) < 1e-4
            qg = ctx.read_quantized(f)
            assert np.array_equal(qg, O.pyramid_from_spectrum(Xg, N, is_real, levels))
            dq = np.abs(qg.astype(np.int16) - g["quantized"][f].astype(np.int16))
            assert dq.max() <= 1 and (dq != 0).mean() <= 1e-3
        for ci, c in enumerate(cls):
            a, p, nan = c.read_audio(nframes)
            for f in range(nframes):
                assert nan[f] == 0
                assert abs(p[f] - g["pwr"][ci, f]) <= 1e-4 * max(abs(g["pwr"][ci, f]), 1e-30)
                if g["clients"][ci][0] == "FM":
                    check_fm(a[f], g["audio"][ci, f], g["baseband"][ci, f], g["baseband_prev"][ci, f], f"client {ci} frame {f}")
                else:
                    assert rel_l2(a[f], g["audio"][ci, f]) < 1e-4
        ctx.dev_free(d)
    finally:
        ctx.close()
