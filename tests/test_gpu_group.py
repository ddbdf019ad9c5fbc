"""psdr_group_* (include/psdr.h): SURVEY 8e from C - one process, n GPUs, the batch exchanged through RCCL called
directly (phantomsdr_amd/csrc/group.hip).  Whatever the sharding, every client's audio must be the SAME BITS as on a
single plain context (the same kernels run on the same spectra; only where they run differs).

* one device with PSDR_SHARD_FORCE_COMM: the whole group path including librccl.so's dlopen, ncclCommInitAll and the
  collectives (a broadcast to oneself; band: pack + ncclSend / ncclRecv to oneself + demodulation from the received
  band buffer) - what a single-GPU box can exercise of RCCL;
* n RANKS ON ONE DEVICE with PSDR_SHARD_PEER_COPY (the peers pull with device copies instead of RCCL): the multi-rank
  logic itself - round-robin / band placement, band regions and halos, raw replication, per-rank fetch, migration of a
  client between bands with its state - runs for real on a one-GPU box;
* two and more physical devices over RCCL: skipped unless the box has them."""
import ctypes as C

import numpy as np
import pytest

from helpers import quantize_raw, synth_stream
from test_gpu_parity import levels_for

pytestmark = pytest.mark.gpu


def _ndev():
    hip = C.CDLL("libamdhip64.so")
    n = C.c_int(0)
    return n.value if hip.hipGetDeviceCount(C.byref(n)) == 0 else 0


def _clients(N, R, n, count):
    rng = np.random.default_rng(5)
    out = []
    for i in range(count):
        mode = ("USB", "LSB", "AM", "FM")[i % 4]
        # spread over the whole spectrum so that every band / every rank gets some
        m = int((i + 0.5) * R / count + rng.integers(-50, 50))
        m = min(max(m, 200), R - 200)
        w = 60
        l, r = (m, m + w) if mode == "USB" else (m - w, m) if mode == "LSB" else (m - w, m + w)
        out.append((mode, l, float(m) + (0.5 if i % 3 == 0 else 0.0), r))
    return out


def _run_plain(N, is_real, n, F, nb, raw, clients, levels):
    from phantomsdr_amd import AudioClient, Context
    ctx = Context(N, is_real, levels, additional_size=n, audio_fft_size=n, input_format="s16", max_batch=F,
                  max_clients=len(clients), max_waterfall_clients=2)
    try:
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        gcl = []
        for mode, l, m, r in clients:
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            g.set_audio_range(l, m, r)
            gcl.append(g)
        hb = ctx.half_frame_bytes()
        out = []
        for b in range(nb):
            ctx.process_batch(d, F, offset_bytes=b * F * hb)
            ctx.demod_batch(b * F)
            out.append([g.read_audio(F) for g in gcl])
        spec = ctx.read_spectrum(F - 1)
        q = ctx.read_quantized(F - 1)
        ctx.dev_free(d)
        return out, spec, q
    finally:
        ctx.close()


def _run_group(devices, shard, force, N, is_real, n, F, nb, raw, clients, levels, peer_copy=False):
    from phantomsdr_amd import Group
    g = Group(devices, shard, N, is_real, levels, force_comm=force, peer_copy=peer_copy, additional_size=n, audio_fft_size=n,
              input_format="s16", max_batch=F, max_clients=len(clients), max_waterfall_clients=2)
    try:
        root = g.root
        d = root.dev_alloc(raw.nbytes)
        root.h2d(d, raw)
        gids = [g.client_add(l, m, r, mode) for mode, l, m, r in clients]
        ranks = sorted({g.client_rank(gid) for gid in gids})
        hb = root.half_frame_bytes()
        out = []
        links = []
        for b in range(nb):
            g.step(d, F, b * F, offset_bytes=b * F * hb)
            g.fetch()
            links.append(g.link_stats())
            batch = []
            for gid in gids:
                rows = [g.fetched_audio(gid, f) for f in range(F)]
                batch.append((np.stack([x[0] for x in rows]), np.array([x[1] for x in rows], np.float32),
                              np.array([x[2] for x in rows], np.int32)))
            out.append(batch)
        g.synchronize()
        spec = root.read_spectrum(F - 1)
        q = root.read_quantized(F - 1)
        root.dev_free(d)
        return out, spec, q, ranks, links
    finally:
        g.close()


def _same(a, b, tag):
    for bi, (ba, bb) in enumerate(zip(a, b)):
        for ci, (ca, cb) in enumerate(zip(ba, bb)):
            for name, u, v in zip(("audio", "pwr", "nan"), ca, cb):
                u, v = np.asarray(u), np.asarray(v)
                assert np.array_equal(u.view(np.uint32) if u.dtype == np.float32 else u, v.view(np.uint32) if v.dtype == np.float32 else v), \
                    f"{tag}: batch {bi} client {ci} {name} differs"


CASES = [(1 << 16, 0, 248), (1 << 20, 0, 360), (1 << 17, 1, 248)]


@pytest.mark.parametrize("shard", ["clients", "raw", "band"])
@pytest.mark.parametrize("N,is_real,n", CASES)
def test_single_device_group_with_forced_rccl_matches_a_plain_context(shard, N, is_real, n):
    F, nb = 4, 3
    R = N // 2 if is_real else N
    levels = levels_for(R)
    x = synth_stream((nb * F + 1) * (N // 2), bool(is_real), seed=21, fft_size=N)
    raw = quantize_raw(x, "s16", bool(is_real))
    clients = _clients(N, R, n, 12)
    ref, spec_r, q_r = _run_plain(N, is_real, n, F, nb, raw, clients, levels)
    got, spec_g, q_g, ranks, links = _run_group([0], shard, True, N, is_real, n, F, nb, raw, clients, levels)
    _same(ref, got, f"{shard} N=2^{N.bit_length() - 1}")
    assert np.array_equal(spec_r.view(np.uint32), spec_g.view(np.uint32)) and np.array_equal(q_r, q_g)
    assert ranks == [0]
    # bytes crossed the (self) link, the exchange took time: RCCL ran (band: the packed band sent to and received from oneself)
    assert links[-1][0] > 0 and links[-1][1] > 0, links


@pytest.mark.parametrize("ndev", [2, 4, 8])
@pytest.mark.parametrize("shard", ["clients", "raw", "band"])
@pytest.mark.parametrize("N,is_real,n", CASES)
def test_n_ranks_on_one_device_with_peer_copies_match_a_plain_context(shard, N, is_real, n, ndev):
    """n contexts on device 0, the exchange by device copies (PSDR_SHARD_PEER_COPY): every rank serves clients, and every
    client's audio is the bits of a plain context - band sharding through the root's band regions + halos (2^20 IQ) or
    the pack (2^16 IQ, real input), raw through n forward transforms"""
    F, nb = 4, 3
    R = N // 2 if is_real else N
    levels = levels_for(R)
    x = synth_stream((nb * F + 1) * (N // 2), bool(is_real), seed=21, fft_size=N)
    raw = quantize_raw(x, "s16", bool(is_real))
    clients = _clients(N, R, n, 24)
    ref, spec_r, q_r = _run_plain(N, is_real, n, F, nb, raw, clients, levels)
    got, spec_g, q_g, ranks, links = _run_group([0] * ndev, shard, False, N, is_real, n, F, nb, raw, clients, levels, peer_copy=True)
    _same(ref, got, f"{shard} x{ndev} ranks on one device N=2^{N.bit_length() - 1}")
    assert np.array_equal(spec_r.view(np.uint32), spec_g.view(np.uint32)) and np.array_equal(q_r, q_g)
    assert ranks == list(range(ndev)), "every rank serves clients"
    assert links[-1][0] > 0 and links[-1][1] > 0


def _run_group_unsynced(devices, shard, force, peer_copy, serial, N, is_real, n, F, steps, raw, clients, levels):
    """`steps` batches enqueued back to back - no fetch, no synchronisation in between - then the LAST batch's results"""
    from phantomsdr_amd import Group
    g = Group(devices, shard, N, is_real, levels, force_comm=force, peer_copy=peer_copy, serial=serial, additional_size=n, audio_fft_size=n,
              input_format="s16", max_batch=F, max_clients=len(clients), max_waterfall_clients=2)
    try:
        root = g.root
        d = root.dev_alloc(raw.nbytes)
        root.h2d(d, raw)
        gids = [g.client_add(l, m, r, mode) for mode, l, m, r in clients]
        hb = root.half_frame_bytes()
        for b in range(steps):
            g.step(d, F, b * F, offset_bytes=b * F * hb)
        g.fetch()
        g.synchronize()
        batch = []
        for gid in gids:
            rows = [g.fetched_audio(gid, f) for f in range(F)]
            batch.append((np.stack([x[0] for x in rows]), np.array([x[1] for x in rows], np.float32), np.array([x[2] for x in rows], np.int32)))
        spec = root.read_spectrum(F - 1)
        root.dev_free(d)
        return batch, spec
    finally:
        g.close()


@pytest.mark.parametrize("how,ndev,shard", [("peer_copy", 8, "clients"), ("peer_copy", 4, "band"), ("rccl_forced", 1, "clients")])
def test_overlapped_exchange_is_bit_identical_to_the_serial_step(how, ndev, shard):
    """The exchange of batch b runs beside the root's transform of batch b + 1 (the root alternates its result sets, its
    side of the collective sits on an exchange stream, a set is overwritten only after its exchange: group.hip).  k batches
    enqueued WITHOUT any synchronisation in between, k = 1 .. 5: the k-th batch's audio on every rank, and the root's
    spectrum, are the bits of the serial schedule (PSDR_SHARD_SERIAL) and of a plain context that steps batch by batch -
    through n ranks on one device with peer copies (clients; band regions of the 2^20-point IQ second pass) and through
    RCCL forced on one rank."""
    N, is_real, n, F, nb = 1 << 20, 0, 360, 3, 5
    R = N
    levels = levels_for(R)
    x = synth_stream((nb * F + 1) * (N // 2), False, seed=41, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    clients = _clients(N, R, n, 16)
    ref, _, _ = _run_plain(N, is_real, n, F, nb, raw, clients, levels)
    for k in range(1, nb + 1):
        res = {}
        for serial in (True, False):
            res[serial] = _run_group_unsynced([0] * ndev, shard, how == "rccl_forced", how == "peer_copy", serial, N, is_real, n, F, k, raw,
                                              clients, levels)
        _same([ref[k - 1]], [res[True][0]], f"serial {how} x{ndev} {shard}, batch {k} of {k} unsynchronised")
        _same([ref[k - 1]], [res[False][0]], f"overlapped {how} x{ndev} {shard}, batch {k} of {k} unsynchronised")
        assert np.array_equal(res[True][1].view(np.uint32), res[False][1].view(np.uint32))


@pytest.mark.parametrize("N,is_real,n", [(1 << 16, 0, 248), (1 << 20, 0, 360)])
def test_band_migration_keeps_the_gid_and_the_demodulation_state(N, is_real, n):
    """Band sharding, 4 ranks (on one device): clients retune across band edges between batches - the gid stays, the rank
    changes, and because the overlap-add tails / FM's last sample travel with the client every sample after the move is
    the plain context's (which retunes the same clients at the same batch boundaries).  The batch in flight when the move
    happens reads as PSDR_ERR_NO_DATA through the group until the new rank has demodulated (documented in psdr.h)."""
    from phantomsdr_amd import AudioClient, Context, Group, PsdrError
    F, nb, ndev = 3, 5, 4
    R = N
    levels = levels_for(R)
    x = synth_stream((nb * F + 1) * (N // 2), False, seed=33, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    per = R // ndev
    # (mode, window before, window after): across one edge, across two, back into band 0, and one that stays
    def win(mode, m, w=60):
        return (m, float(m), m + w) if mode == "USB" else (m - w, float(m) + 0.5, m) if mode == "LSB" else (m - w, float(m), m + w)
    plan = [("USB", per - 500, per + 700), ("LSB", per // 2, 3 * per - 90), ("AM", 3 * per + 4000, 300), ("FM", 2 * per + 100, 2 * per + 9000),
            ("FM", per + 50, 3 * per + 77)]
    move_at = 2  # the retune lands between batch 1 and batch 2
    ctx = Context(N, False, levels, additional_size=n, audio_fft_size=n, input_format="s16", max_batch=F, max_clients=len(plan), max_waterfall_clients=1)
    try:
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        pcl = []
        for mode, m0, _ in plan:
            c = AudioClient(ctx)
            c.set_audio_demodulation(mode)
            c.set_audio_range(*win(mode, m0))
            pcl.append(c)
        hb = ctx.half_frame_bytes()
        ref = []
        for b in range(nb):
            if b == move_at:
                for c, (mode, _, m1) in zip(pcl, plan):
                    c.set_audio_range(*win(mode, m1))
            ctx.process_batch(d, F, offset_bytes=b * F * hb)
            ctx.demod_batch(b * F)
            ref.append([c.read_audio(F) for c in pcl])
        ctx.dev_free(d)
    finally:
        ctx.close()
    g = Group([0] * ndev, "band", N, False, levels, peer_copy=True, additional_size=n, audio_fft_size=n, input_format="s16", max_batch=F,
              max_clients=len(plan), max_waterfall_clients=1)
    try:
        root = g.root
        d = root.dev_alloc(raw.nbytes)
        root.h2d(d, raw)
        gids = [g.client_add(*win(mode, m0), mode) for mode, m0, _ in plan]
        before = [g.client_rank(gid) for gid in gids]
        hb = root.half_frame_bytes()
        for b in range(nb):
            if b == move_at:
                for gid, (mode, _, m1) in zip(gids, plan):
                    assert g.client_set_audio_range(gid, *win(mode, m1)) == gid
                after = [g.client_rank(gid) for gid in gids]
                assert [a != b_ for a, b_ in zip(after, before)] == [True, True, True, False, True], (before, after)
                # the batch demodulated before the move is gone for the clients that moved
                for gid, moved in zip(gids, [True, True, True, False, True]):
                    if moved:
                        with pytest.raises(PsdrError):
                            g.fetched_audio(gid, 0)
            g.step(d, F, b * F, offset_bytes=b * F * hb)
            g.fetch()
            for ci, gid in enumerate(gids):
                for f in range(F):
                    a, pw, nan = g.fetched_audio(gid, f)
                    ra, rp, rn = ref[b][ci][0][f], ref[b][ci][1][f], ref[b][ci][2][f]
                    assert np.array_equal(a.view(np.uint32), ra.view(np.uint32)), f"batch {b} client {ci} frame {f}: audio differs"
                    assert np.float32(pw).view(np.uint32) == np.float32(rp).view(np.uint32) and nan == rn
        g.synchronize()
        root.dev_free(d)
    finally:
        g.close()


@pytest.mark.skipif(_ndev() < 2, reason="needs two HIP devices")
@pytest.mark.parametrize("ndev", [2, 4, 8])
@pytest.mark.parametrize("shard", ["clients", "raw", "band"])
@pytest.mark.parametrize("N,is_real,n", CASES)
def test_multi_device_group_matches_a_plain_context(shard, N, is_real, n, ndev):
    if _ndev() < ndev:
        pytest.skip(f"needs {ndev} HIP devices")
    F, nb = 4, 3
    R = N // 2 if is_real else N
    levels = levels_for(R)
    x = synth_stream((nb * F + 1) * (N // 2), bool(is_real), seed=21, fft_size=N)
    raw = quantize_raw(x, "s16", bool(is_real))
    clients = _clients(N, R, n, 24)
    ref, spec_r, q_r = _run_plain(N, is_real, n, F, nb, raw, clients, levels)
    got, spec_g, q_g, ranks, links = _run_group(list(range(ndev)), shard, False, N, is_real, n, F, nb, raw, clients, levels)
    _same(ref, got, f"{shard} x{ndev} N=2^{N.bit_length() - 1}")
    assert np.array_equal(spec_r.view(np.uint32), spec_g.view(np.uint32)) and np.array_equal(q_r, q_g)
    assert ranks == list(range(ndev)), "every device serves clients"
    assert links[-1][0] > 0 and links[-1][1] > 0


def test_group_argument_errors():
    from phantomsdr_amd import Group, PsdrError
    with pytest.raises(PsdrError) as e:
        Group([0, 0], "clients", 1 << 16, False, 7, audio_fft_size=248, additional_size=248)
    assert e.value.code == -1 and "twice" in str(e.value)
    with pytest.raises(PsdrError) as e:  # RCCL forced and no RCCL at all exclude each other
        Group([0], "clients", 1 << 16, False, 7, force_comm=True, peer_copy=True, audio_fft_size=248, additional_size=248)
    assert e.value.code == -1
    if _ndev() >= 3:
        with pytest.raises(PsdrError) as e:
            Group([0, 1, 2], "band", 1 << 16, False, 7, audio_fft_size=248, additional_size=248)
        assert e.value.code == -1
    g = Group([0], "band", 1 << 16, False, 7, audio_fft_size=248, additional_size=248, max_clients=4)
    try:
        gid = g.client_add(100, 100.0, 160, "USB")
        assert g.client_rank(gid) == 0 and g.client_rank(gid + 1) == -1
        assert g.client_set_audio_range(gid, 60000, 60000.0, 60060) == gid  # one band: nothing to migrate to
        assert g.client_rank(gid) == 0
    finally:
        g.close()


@pytest.mark.parametrize("shard", [0, 1, 2])
def test_plain_c_caller_of_the_group_matches_the_oracle(shard):
    """examples/group_demo.c: SURVEY 8e from plain C (one process, psdr_group_*; here one device with the collectives
    forced).  Its dumped per-client audio against the oracle on its dumped stream."""
    import atexit
    import os
    import shutil
    import subprocess
    import tempfile
    from conftest import ROOT
    from helpers import rel_l2
    from oracle import oracle as O
    d = tempfile.mkdtemp(prefix="psdr_grp_")
    atexit.register(shutil.rmtree, d, ignore_errors=True)
    exe = os.path.join(d, "group_demo")
    lib = os.path.join(ROOT, "phantomsdr_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "group_demo.c"),
                           "-L" + lib, "-lpsdr_hip", "-lm", "-Wl,-rpath," + lib, "-o", exe])
    r = subprocess.run([exe, "1", str(shard), d], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "group demo ok" in r.stdout, r.stdout + r.stderr
    N, F, NB, NCL, n, levels = 1 << 16, 4, 3, 8, 248, 7
    raw = np.fromfile(os.path.join(d, "raw.bin"), np.int16)
    halves = O.convert(raw, "s16").view(np.complex64).reshape(NB * F + 1, N // 2)
    fo = O.FFT(N, False, levels, 0, n)
    ocl = []
    for i in range(NCL):
        k = int((i + 0.5) * N / NCL) - N // 2
        c = (k - (N // 2 + 1)) % N
        o = O.AudioClient(False, n, 12000, N)
        o.set_audio_demodulation("USB")
        o.set_audio_range(c, float(c), c + 60)
        ocl.append(o)
    got = [np.fromfile(os.path.join(d, f"client{i}.bin"), np.float32).reshape(NB * F, n // 2) for i in range(NCL)]
    for f in range(NB * F):
        fo.load(halves[f], halves[f + 1])
        fo.execute()
        for i, o in enumerate(ocl):
            a_o, _, _, dropped = o.send_audio(fo.output(), f, fft=fo)
            assert not dropped and rel_l2(got[i][f], a_o) < 1e-4, (shard, i, f, rel_l2(got[i][f], a_o))
