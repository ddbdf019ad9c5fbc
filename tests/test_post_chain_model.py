"""The index algebra of the post chain's one-kernel AGC (phantomsdr_amd/csrc/postchain.h: k_pc_cm, k_pc_cscan, k_pc_agc) restated in
numpy against the brute-force look-ahead window - what the reference's monotonic deque computes (src/utils/audioprocessing.cpp:
18-38: the maximum of |x| over the last L samples).  No GPU: the kernels' own bit-exactness is tests/test_gpu_parity.py's business;
this pins the decomposition they rest on, including that values past a stream's end never reach a peak."""
import numpy as np
import pytest


def brute(v, L, T):
    # rows 0 .. L-2 are history, sample t sits in row L-1+t; output step t sees rows t .. t+L-1
    return np.array([np.abs(v[t:t + L]).max() for t in range(T)])


def chunked(v, L, T, rng):
    """V1 as the kernels see it: one leading pad float (vo = 1: sample 0's row is float L = chunk L/16's first), chunks of 16
    floats, CM = chunk maxima, CP / CS = prefix / suffix maxima of CM inside blocks of W = L/16 - 1 chunks."""
    assert L % 16 == 0 and L >= 32
    LC, W = L // 16, L // 16 - 1
    n = len(v)
    idx = np.zeros(1 + n + 16 * (W + 8))
    idx[1:1 + n] = np.abs(v)
    idx[1 + n:] = rng.random(len(idx) - 1 - n) * 1e6  # what lies past the stream's end: stale rows of an earlier batch
    nch = len(idx) // 16
    CM = idx[:nch * 16].reshape(nch, 16).max(1)
    CP, CS = np.zeros(nch), np.zeros(nch)
    for k0 in range(0, nch, W):
        k1 = min(k0 + W, nch)
        CP[k0:k1] = np.maximum.accumulate(CM[k0:k1])
        CS[k0:k1] = np.maximum.accumulate(CM[k0:k1][::-1])[::-1]
    out = np.zeros(T)
    for b in range((T + 15) // 16):
        near, far = idx[16 * b:16 * b + 16], idx[16 * (b + LC):16 * (b + LC) + 16]
        mid = max(CS[b + 1], CP[b + W])
        sfx = np.zeros(17)  # sfx[p] = max near[p .. 15], sfx[16] = 0 (empty)
        for p in range(15, 0, -1):
            sfx[p] = max(sfx[p + 1], near[p])
        pfx = np.maximum.accumulate(far)
        for i in range(16):
            if 16 * b + i < T:
                out[16 * b + i] = max(sfx[i + 1], mid, pfx[i])
    return out


@pytest.mark.parametrize("L,T", [(32, 100), (48, 7), (2400, 5000), (2400, 16 * 31 + 4), (1200, 3000), (2400, 1260), (9600, 4092)])
def test_chunked_look_ahead_peak_is_the_sliding_window_maximum(L, T):
    rng = np.random.default_rng(L + T)
    v = rng.standard_normal(L - 1 + T) * rng.random(L - 1 + T) ** 4
    assert np.array_equal(brute(v, L, T), chunked(v, L, T, rng))


def test_stream_position_to_frame_by_one_multiplication():
    """k_pc_agc finds the frame of a stream position with t / h = umulhi(t, ceil(2^32 / h)) - exact while t * h < 2^32
    (psdr_set_post_chain checks the batch against that bound)."""
    for h in (16, 124, 180, 360, 5034):
        magic = ((1 << 32) + h - 1) // h
        tmax = min((1 << 32) // h - 1, 5_000_000)
        t = np.unique(np.concatenate([np.arange(0, min(tmax, 200_000)), np.random.default_rng(h).integers(0, tmax, 200_000), [tmax]])).astype(np.uint64)
        assert np.array_equal((t * np.uint64(magic)) >> np.uint64(32), t // np.uint64(h)), h
