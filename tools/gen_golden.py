#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference (PhantomSDR) has no tests and no golden vectors, and its path does not
compile in this image (fftw3.h / boost / websocketpp are absent), so these fixtures are
outputs of oracle/psdr_oracle.c — whose helpers are pinned bit-exactly against the
reference's own compiled dsp.cpp/audioprocessing.cpp (oracle/_ref) and whose DFTs are pinned
against float64 numpy.  They freeze the oracle's behaviour (any drift fails
tests/test_golden.py) and travel to the GPU box as data.

  python tools/gen_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import quantize_raw, synth_stream  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

CASES = [
    # name, N, is_real, fmt, n (audio fft), levels, nframes, seed
    ("iq4096_s16_n60", 4096, 0, "s16", 60, 3, 6, 101),
    ("iq4096_u8_n8", 4096, 0, "u8", 8, 3, 4, 102),
    ("real8192_s16_n60", 8192, 1, "s16", 60, 3, 6, 103),
    ("iq16384_f32_n248", 16384, 0, "f32", 248, 5, 4, 104),
]


def clients_for(N, is_real, n):
    R = N // 2 if is_real else N
    am = int((0.11 * N) if is_real else ((0.11 * N - (N // 2 + 1)) % N))
    w = max(2, n // 4)
    h = max(2, n // 2 - 1)
    cl = [("USB", am, float(am), am + w), ("USB", am + 1, am + 1.5, am + 1 + w),
          ("LSB", am - w, float(am), am), ("AM", am - h, float(am), am + h),
          ("FM", am - h, am + 0.25, am + h), ("USB", 0, 0.0, w), ("LSB", R - 1 - w, float(R - 1), R - 1)]
    if not is_real:
        dc = N // 2 - 1
        cl.append(("AM", dc - h, float(dc), dc + h))
    return cl


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, N, is_real, fmt, n, levels, nframes, seed in CASES:
        sigma = 2.0 ** -5 if fmt == "u8" else 2.0 ** -9
        x = synth_stream((nframes + 1) * (N // 2), is_real, seed=seed, sigma=sigma, fft_size=N)
        raw = quantize_raw(x, fmt, is_real)
        conv = O.convert(raw, fmt)
        halves = (conv if is_real else conv.view(np.complex64)).reshape(nframes + 1, N // 2)
        R = N // 2 if is_real else N
        fo = O.FFT(N, is_real, levels, 0, n)
        cl = clients_for(N, is_real, n)
        ocl = []
        for mode, l, m, r in cl:
            c = O.AudioClient(is_real, n, 12000, R)
            c.set_audio_demodulation(mode)
            c.set_audio_range(l, m, r)
            ocl.append(c)
        nb = N // 2 + 1 if is_real else N + n
        spec = np.zeros((nframes, nb), np.complex64)
        quant = np.zeros((nframes, sum(R >> i for i in range(levels))), np.int8)
        audio = np.zeros((len(cl), nframes, n // 2), np.float32)
        pwr = np.zeros((len(cl), nframes), np.float32)
        pcm = np.zeros((len(cl), nframes, n // 2), np.int32)
        # the complex baseband the AM/FM outputs are taken from (and the sample FM's first output pairs with):
        # the tests condition the FM bound on it (tests/helpers.py fm_tolerance)
        bb = np.zeros((len(cl), nframes, n // 2), np.complex64)
        bb_prev = np.zeros((len(cl), nframes), np.complex64)
        for f in range(nframes):
            fo.load(halves[f], halves[f + 1])
            fo.execute()
            spec[f] = fo.output()
            quant[f] = fo.quantized()
            for ci, c in enumerate(ocl):
                a, p, pc, _ = c.send_audio(spec[f], f, fft=fo, post=True)
                audio[ci, f], pwr[ci, f], pcm[ci, f] = a, p, pc
                bb[ci, f], bb_prev[ci, f] = c.baseband()[: n // 2], c.bb_prev
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"), raw=raw, fmt=fmt, N=N, is_real=is_real, n=n,
            levels=levels, spectrum=spec, quantized=quant, audio=audio, pwr=pwr, pcm=pcm, baseband=bb,
            baseband_prev=bb_prev,
            client_modes=np.array([c[0] for c in cl]), client_l=np.array([c[1] for c in cl]),
            client_m=np.array([c[2] for c in cl]), client_r=np.array([c[3] for c in cl]))
        print(name, os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
