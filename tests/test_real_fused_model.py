"""CPU model of the fused real-input pass 2 (phantomsdr_amd/csrc/fft_pass.h, k_fft_pass2_real):
the index algebra the kernel relies on, restated in numpy at small sizes and checked against
numpy.fft.rfft.  This is host-side logic (no GPU): it pins

  * the row pairing of pass 1's PAIR mode: pass-2 tile g = rows 8g..8g+7 + their mirrors M1-8g-p stored
    as conj(Y)*W_M2^{n2}, row M1/2 beside row 0;
  * the thread-local untangle X[k] = (a+b)/2 + W_N^k(-i)(a-b)/2, X[M-k] = conj((a+b)/2 - ...), with
    a = Z[k], b = conj(Z[M-k]) = the mirror row's plain forward transform at the same output index;
  * the self-paired rows 0 and M1/2 (couple 0 of tile 0) and the un-normalised bin N/2
    (src/fft_impl.cpp:156-160 never visits it);
  * the line layout of the spectrum (SpecLayout): the two bins a couple produces together share a
    128-byte line, so one store instruction writes whole lines;
  * the octet pyramid bookkeeping: low octets complete per tile, high octets through the carried row of
    the chain (decreasing g), the seam buffers at segment boundaries and the ring closure through row
    M1/2, and the record order (RecMap mode 2).
"""
import numpy as np
import pytest


def spec_layout_pos(k, M1, M2, CP=8):
    """SpecLayout::pos (quantize.h): lines of 2 CP bins [low octet of column c | mirror octet of column L-1-c],
    tile-major (line (g, c) = g*L + c); CP = couples per tile: 8 (128-byte lines), or 4 with 2048-point rows"""
    c1, c2 = k % M1, k // M1
    if c1 < M1 // 2:
        return ((c1 // CP) * M2 + c2) * (2 * CP) + c1 % CP
    hp = M1 - 1 if c1 == M1 // 2 else c1 - 1
    g = (M1 - 1 - hp) // CP
    return (g * M2 + (M2 - 1 - c2)) * (2 * CP) + CP + hp % CP


def recmap2_pos(o, M1, M2, CP=8):
    tpr = M1 // CP
    row, gc = o // tpr, o % tpr
    side = 1 if gc >= tpr // 2 else 0
    tl = tpr - 1 - gc if side else gc
    return ((tl * M2) + row) * 2 + side


def pass1_pair(z, M1, M2, CP=8):
    """Y slots [M1][M2] as pass 1 stores them in PAIR mode"""
    M = M1 * M2
    zz = z.reshape(M1, M2)                       # z[M2*n1 + n2]
    col = np.fft.fft(zz, axis=0)                 # over n1 -> [c1][n2]
    n2 = np.arange(M2)
    Y = col * np.exp(-2j * np.pi * np.outer(np.arange(M1), n2) / M)
    slots = np.zeros((M1, M2), complex)
    cW = np.exp(-2j * np.pi * n2 / M2)
    for c1 in range(M1):
        if c1 < M1 // 2:
            slots[2 * CP * (c1 // CP) + c1 % CP] = Y[c1]
        elif c1 == M1 // 2:
            slots[CP] = Y[c1]                    # natural form, beside row 0 (couple 0 of tile 0)
        else:
            m = M1 - c1
            slots[2 * CP * (m // CP) + CP + m % CP] = np.conj(Y[c1]) * cW
    return slots


def untangle_pair(a, b, w, h):
    s, d = a + b, a - b
    wo = w * (-1j * d)
    return (s + wo) * h, np.conj((s - wo) * h)


def seg_plan(G, nframes, num_cus=256, seg_len_env=0, static_tiles=False):
    """forward.hip: seg_plan_counts / seg_plan.  Returns (table, handoff, nseam): table[sg] = (frame, first tile, tiles,
    segment above, carry-in through memory)."""
    def pow2_floor(v):
        sl = 1
        while sl * 2 <= v and sl * 2 <= G:
            sl *= 2
        return sl

    def uniform_len():   # forward.hip: real_seg_len
        if seg_len_env > 0:
            return pow2_floor(seg_len_env)
        W = max(num_cus, 1)
        tiles = G * nframes
        want = tiles // (2 * W)
        per_wg = -(-tiles // W)
        if per_wg <= 16:                     # one static segment per work-group
            sl = 1
            while sl < per_wg and sl * 2 <= G:
                sl *= 2
            return sl
        if 2 * W * want == tiles and want & (want - 1) == 0 and want <= G:
            return want                      # two static segments per work-group
        return pow2_floor(tiles // (5 * W))  # five to ten per work-group: the tickets have something to balance with
    want = nframes > num_cus                 # the hand-off plan: more than one frame per work-group
    handoff = seg_len_env <= 0 and not static_tiles and G >= 16 and want
    tab = []
    if handoff:
        lens = [G // 4] * 3
        l = G // 8
        while l >= 1:
            lens.append(l)
            l //= 2
        lens.append(1)
        g = G - 1
        for lv, ln in enumerate(lens):
            for f in range(nframes):
                tab.append((f, g, ln, ((lv - 1) if lv else len(lens) - 1) * nframes + f, lv > 0))
            g -= ln
        return tab, True, nframes
    sl = uniform_len()
    S = G // sl
    for f in range(nframes):
        for si in range(S):
            tab.append((f, (si + 1) * sl - 1, sl, f * S + (si + 1) % S, False))
    return tab, False, len(tab)


def fused_pass2(slots, M1, M2, seg_len, CP=8):
    """uniform segments of seg_len tiles, every one with a seam (small batches, PSDR_SEG_LEN)"""
    G = M1 // (2 * CP)
    S = G // seg_len
    return fused_pass2_segments(slots, M1, M2, [((si + 1) * seg_len - 1, seg_len, (si + 1) % S, False) for si in range(S)], CP)


def fused_pass2_segments(slots, M1, M2, segs, CP=8):
    """One frame.  segs[i] = (first tile, tiles, index of the segment above, carry-in through memory); processed in the
    order given (a segment with a carry-in through memory needs its predecessor's carry-out: the order must provide it,
    as the ticket order of the kernel does).  Returns (permuted spectrum [M+1], records dict pos -> 8 powers)."""
    M, N = M1 * M2, 2 * M1 * M2
    h = 0.5 / N
    X = np.zeros(M + 1, complex)
    rec, seamP, seamC = {}, {}, {}
    c2 = np.arange(M2)
    for si, (g_first, seg_len, above, carry_mem) in enumerate(segs):
        carry = seamC[above].copy() if carry_mem else None   # (KeyError: the plan handed the segments out in a wrong order)
        for j in range(seg_len):
            g = g_first - j
            LN = 2 * CP
            low = np.zeros((M2, CP))
            high = np.zeros((M2, CP))
            carry_w = np.zeros(M2)
            for p in range(CP):
                a = np.fft.fft(slots[LN * g + p])
                b = np.fft.fft(slots[LN * g + CP + p])
                c1 = CP * g + p
                if g == 0 and p == 0:
                    # row 0 <-> itself at column (M2-c2) % M2; row M1/2 <-> itself at column M2-1-c2
                    w = np.exp(-2j * np.pi * (M1 * c2) / N)
                    xk, _ = untangle_pair(a, np.conj(a[(M2 - c2) % M2]), w, h)
                    X[LN * c2] = xk                       # line (0, c2), bin 0
                    low[:, 0] = np.abs(xk) ** 2
                    X[M] = a[0].real - a[0].imag
                    w = np.exp(-2j * np.pi * (M1 // 2 + M1 * c2) / N)
                    xk, _ = untangle_pair(b, np.conj(b[M2 - 1 - c2]), w, h)
                    X[LN * (M2 - 1 - c2) + LN - 1] = xk   # row M1/2 closes tile 0's mirror octet
                    seamC[si] = np.abs(xk) ** 2
                    continue
                w = np.exp(-2j * np.pi * (c1 + M1 * c2) / N)
                xk, xm = untangle_pair(a, b, w, h)
                cm = M2 - 1 - c2
                X[(g * M2 + c2) * LN + p] = xk            # both halves of line (g, c2)
                X[(g * M2 + c2) * LN + LN - 1 - p] = xm
                low[:, p] = np.abs(xk) ** 2
                if p:
                    high[cm, CP - p] = np.abs(xm) ** 2
                else:
                    carry_w[cm] = np.abs(xm) ** 2
                    if j == seg_len - 1:
                        seamC[si] = carry_w.copy()      # (hand-off plan: published behind segflag[si])
            for c in range(M2):
                rec[(g * M2 + c) * 2] = low[c].copy()
                if carry is None:
                    seamP[si] = high.copy() if c == 0 else seamP[si]
                else:
                    pw = high[c].copy()
                    pw[0] = carry[c]
                    rec[(g * M2 + c) * 2 + 1] = pw
            carry = carry_w
    for si, (g, _, above, carry_mem) in enumerate(segs):   # k_real_seam: the segments without a carry-in
        if carry_mem:
            continue
        for c in range(M2):
            pw = seamP[si][c].copy()
            pw[0] = seamC[above][c]
            rec[(g * M2 + c) * 2 + 1] = pw
    return X, rec


@pytest.mark.parametrize("M1,M2,seg_len,CP", [(32, 16, 1, 8), (32, 16, 2, 8), (64, 8, 2, 8), (64, 8, 4, 8), (16, 16, 1, 8),
                                              (32, 16, 1, 4), (32, 16, 2, 4), (32, 64, 4, 4), (16, 32, 2, 4), (8, 16, 1, 4)])
def test_fused_real_pass2_model_matches_rfft(M1, M2, seg_len, CP):
    """CP = 8: tiles of eight (row, mirror row) couples, octet records (1024-point rows); CP = 4: four couples, quartet
    records, lines of 8 bins (2048-point rows: 2^22-point real frames split 1024 x 2048)"""
    M, N = M1 * M2, 2 * M1 * M2
    rng = np.random.default_rng(M1 + seg_len)
    x = rng.standard_normal(N)
    z = x[0::2] + 1j * x[1::2]
    Xp, rec = fused_pass2(pass1_pair(z, M1, M2, CP), M1, M2, seg_len, CP)
    ref = np.fft.rfft(x) / N
    ref[M] *= N                                   # bin N/2 stays un-normalised
    pos = np.array([spec_layout_pos(int(k), M1, M2, CP) for k in range(M)])
    assert sorted(pos) == list(range(M))          # a permutation: every line written exactly once
    got = np.concatenate([Xp[pos], Xp[M:]])
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max() * N
    # octet (quartet) records in true k order through RecMap mode 2
    P = np.abs(ref[:M]) ** 2
    assert len(rec) == M // CP
    for o in range(M // CP):
        assert np.allclose(rec[recmap2_pos(o, M1, M2, CP)], P[CP * o: CP * o + CP], rtol=1e-9, atol=0), o


@pytest.mark.parametrize("G,nframes", [(64, 512), (128, 512), (64, 640), (16, 600), (64, 513), (64, 511), (64, 256), (64, 1), (128, 7), (64, 320), (64, 160), (128, 160),
                                      (64, 128), (64, 192), (64, 96), (64, 384), (64, 257)])
def test_segment_plan_partitions_frames_and_orders_the_hand_offs(G, nframes):
    """the table k_fft_pass2_real walks (forward.hip, seg_plan): every tile of every frame in exactly one segment; the
    segments without a carry-in first (k_real_seam's grid); in hand-off mode a segment's predecessor is the segment of
    the SAME frame that ends one tile above it, handed out exactly nframes tickets earlier, and only a frame's top
    segment needs a seam (the ring closes through tile 0's row M1/2)"""
    tab, handoff, nseam = seg_plan(G, nframes)
    assert handoff == {512: True, 640: True, 600: True, 513: True, 511: True, 256: False, 1: False, 7: False, 320: True,
                       160: False, 128: False, 192: False, 96: False, 384: True, 257: True}[nframes]
    seen = np.zeros((nframes, G), int)
    for f, g0, ln, above, mem in tab:
        assert ln >= 1 and g0 - ln + 1 >= 0
        seen[f, g0 - ln + 1: g0 + 1] += 1
    assert (seen == 1).all()
    assert all(not t[4] for t in tab[:nseam]) and all(t[4] for t in tab[nseam:])
    for sg, (f, g0, ln, above, mem) in enumerate(tab):
        fa, ga, la, _, _ = tab[above]
        assert fa == f
        if g0 == G - 1:
            assert ga - la + 1 == 0 and not mem      # the top segment: row M1/2 from the segment that holds tile 0
        else:
            assert ga - la + 1 == g0 + 1             # the segment above ends one tile above this one's first
        if mem:
            assert above == sg - nframes             # the same frame one level up: >= 2 * grid indices earlier
    if handoff:
        assert nseam == nframes and max(t[2] for t in tab) == G // 4 and tab[-1][2] == 1
        assert len(tab) == nframes * (4 + int(np.log2(G // 8)) + 1)


@pytest.mark.parametrize("M1,M2,CP", [(256, 8, 8), (512, 4, 8), (128, 8, 4), (256, 16, 4)])
def test_fused_real_pass2_model_with_the_hand_off_plan(M1, M2, CP):
    """the hand-off plan's segments of one frame (G/4, G/4, G/4, G/8 ... 1, 1 tiles from the top, carry-in through
    memory for all but the first) through the model: same spectrum and records as numpy.fft.rfft"""
    M, N, G = M1 * M2, 2 * M1 * M2, M1 // (2 * CP)
    tab, handoff, _ = seg_plan(G, 512)
    assert handoff
    segs = [(g0, ln, above // 512, mem) for (f, g0, ln, above, mem) in tab if f == 0]
    rng = np.random.default_rng(M1)
    x = rng.standard_normal(N)
    z = x[0::2] + 1j * x[1::2]
    Xp, rec = fused_pass2_segments(pass1_pair(z, M1, M2, CP), M1, M2, segs, CP)
    ref = np.fft.rfft(x) / N
    ref[M] *= N
    pos = np.array([spec_layout_pos(int(k), M1, M2, CP) for k in range(M)])
    got = np.concatenate([Xp[pos], Xp[M:]])
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max() * N
    P = np.abs(ref[:M]) ** 2
    assert len(rec) == M // CP
    for o in range(M // CP):
        assert np.allclose(rec[recmap2_pos(o, M1, M2, CP)], P[CP * o: CP * o + CP], rtol=1e-9, atol=0), o
