"""world_size-2 gloo test of the multi-GPU orchestration (phantomsdr_amd/distributed.py):
rank 0 produces the spectrum batch, ONE broadcast per batch, every rank demodulates its own
shard of the audio clients (client i -> rank i mod G) with its state kept locally.  The
compute back-end here is the CPU oracle (the HIP back-end needs GPUs); the result must be
identical to a single-process run."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N, NAUD, F, NBATCH, CLIENTS = 4096, 60, 3, 3, 7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clients():
    am = int((0.11 * N - (N // 2 + 1)) % N)
    modes = ["USB", "LSB", "AM", "FM"]
    out = []
    for i in range(CLIENTS):
        mode = modes[i % 4]
        m = am + 3 * i
        if mode == "USB":
            out.append((mode, m, m + 0.5 * (i % 2), m + 14))
        elif mode == "LSB":
            out.append((mode, m - 14, float(m), m))
        else:
            out.append((mode, m - 25, float(m), m + 25))
    return out


class OracleBackend:
    """ShardedRunner back-end on the CPU oracle with torch CPU tensors."""

    def __init__(self, torch, halves, my_clients):
        from oracle import oracle as O
        self.O, self.torch, self.halves = O, torch, halves
        self.fo = O.FFT(N, False, 3, 0, NAUD)
        self.spec = torch.zeros((F, N + NAUD), dtype=torch.complex64)
        self.clients = []
        for mode, l, m, r in my_clients:
            c = O.AudioClient(False, NAUD, 12000, N)
            c.set_audio_demodulation(mode)
            c.set_audio_range(l, m, r)
            self.clients.append(c)
        self.audio = [[] for _ in my_clients]

    def forward(self, i):
        for f in range(F):
            g = i * F + f
            self.fo.load(self.halves[g], self.halves[g + 1])
            self.fo.execute()
            self.spec[f] = self.torch.from_numpy(self.fo.output().copy())

    def spectrum_tensor(self):
        return self.spec

    def demod(self, first_frame_num):
        for f in range(F):
            s = self.spec[f].numpy()
            for ci, c in enumerate(self.clients):
                a, _, _, _ = c.send_audio(s, first_frame_num + f, fft=self.fo)
                self.audio[ci].append(a)


def _halves():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import synth_stream
    x = synth_stream((NBATCH * F + 1) * (N // 2), False, seed=5, fft_size=N).astype(np.complex64)
    return x.reshape(NBATCH * F + 1, N // 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from phantomsdr_amd.distributed import ShardedRunner, assign_clients, gather_audio_to_root
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        allc = _clients()
        mine = assign_clients(len(allc), world)[rank]
        be = OracleBackend(torch, _halves(), [allc[i] for i in mine])
        runner = ShardedRunner(be, dist, rank, world, F)
        for i in range(NBATCH):
            runner.step(i)
        local = [np.stack(a) for a in be.audio]
        merged = gather_audio_to_root(dist, rank, world, mine, local, len(allc))
        if rank == 0:
            q.put((merged, runner.bytes_broadcast, runner.frame_num))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    import torch
    import torch.multiprocessing as mp

    # single-process reference
    sys.path.insert(0, ROOT)
    allc = _clients()
    be = OracleBackend(torch, _halves(), allc)

    class _NoDist:
        pass

    from phantomsdr_amd.distributed import ShardedRunner
    r1 = ShardedRunner(be, _NoDist(), 0, 1, F)
    for i in range(NBATCH):
        r1.step(i)
    ref = [np.stack(a) for a in be.audio]

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, nbytes, frames = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert frames == NBATCH * F
    assert nbytes == NBATCH * F * (N + NAUD) * 8      # one spectrum batch per step
    assert len(merged) == len(ref)
    for a, b in zip(merged, ref):
        assert np.array_equal(a, b)                    # bit-identical to the unsharded run


# ---------------------------------------------------------------------------------------
# the same sharding with the broadcast of batch i overlapping the transform of batch i+1
# ---------------------------------------------------------------------------------------
class OraclePipelinedBackend(OracleBackend):
    def __init__(self, torch, halves, my_clients):
        super().__init__(torch, halves, my_clients)
        self.bufs = torch.zeros((2, F, N + NAUD), dtype=torch.complex64)

    def stage(self, par):
        self.bufs[par].copy_(self.spec)

    def spectrum_tensor(self, par):
        return self.bufs[par]

    def demod(self, first_frame_num, par):
        for f in range(F):
            s = self.bufs[par, f].numpy()
            for ci, c in enumerate(self.clients):
                a, _, _, _ = c.send_audio(s, first_frame_num + f, fft=self.fo)
                self.audio[ci].append(a)


def _pipe_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from phantomsdr_amd.distributed import PipelinedShardedRunner, assign_clients, gather_audio_to_root
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        allc = _clients()
        mine = assign_clients(len(allc), world)[rank]
        be = OraclePipelinedBackend(torch, _halves(), [allc[i] for i in mine])
        runner = PipelinedShardedRunner(be, dist, rank, world, F)
        for i in range(NBATCH):
            runner.step(i)
            assert all(len(a) == i * F for a in be.audio)      # one step late
        runner.flush()
        merged = gather_audio_to_root(dist, rank, world, mine, [np.stack(a) for a in be.audio], len(allc))
        if rank == 0:
            q.put((merged, runner.bytes_broadcast))
    finally:
        dist.destroy_process_group()


def test_two_rank_pipelined_broadcast_matches_single_process():
    import torch
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from phantomsdr_amd.distributed import ShardedRunner
    allc = _clients()
    be = OracleBackend(torch, _halves(), allc)
    r1 = ShardedRunner(be, None, 0, 1, F)
    for i in range(NBATCH):
        r1.step(i)
    ref = [np.stack(a) for a in be.audio]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, nbytes = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert nbytes == NBATCH * F * (N + NAUD) * 8
    for a, b in zip(merged, ref):
        assert np.array_equal(a, b)


# ---------------------------------------------------------------------------------------
# raw-half broadcast (SURVEY 8e variant i): the new half-frames cross the wire, every rank transforms
# ---------------------------------------------------------------------------------------
class OracleRawBackend:
    def __init__(self, torch, halves, my_clients, is_root):
        from oracle import oracle as O
        self.O, self.torch = O, torch
        self.halves = halves if is_root else None
        self.fo = O.FFT(N, False, 3, 0, NAUD)
        self.raw = torch.zeros((F + 1, N // 2), dtype=torch.complex64)
        self.specs = None
        self.clients = []
        for mode, l, m, r in my_clients:
            c = O.AudioClient(False, NAUD, 12000, N)
            c.set_audio_demodulation(mode)
            c.set_audio_range(l, m, r)
            self.clients.append(c)
        self.audio = [[] for _ in my_clients]

    def raw_tensor(self):
        return self.raw

    def load_raw(self, i):
        src = self.torch.from_numpy(self.halves[i * F: i * F + F + 1].copy())
        if i == 0:
            self.raw.copy_(src)
        else:
            self.raw[1:].copy_(src[1:])

    def roll(self):
        self.raw[0].copy_(self.raw[F])

    def forward_local(self):
        self.specs = []
        h = self.raw.numpy()
        for f in range(F):
            self.fo.load(h[f], h[f + 1])
            self.fo.execute()
            self.specs.append(self.fo.output().copy())

    def demod(self, first_frame_num):
        for f in range(F):
            for ci, c in enumerate(self.clients):
                a, _, _, _ = c.send_audio(self.specs[f], first_frame_num + f, fft=self.fo)
                self.audio[ci].append(a)


def _raw_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from phantomsdr_amd.distributed import RawShardedRunner, assign_clients, gather_audio_to_root
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        allc = _clients()
        mine = assign_clients(len(allc), world)[rank]
        be = OracleRawBackend(torch, _halves(), [allc[i] for i in mine], rank == 0)
        runner = RawShardedRunner(be, dist, rank, world, F)
        for i in range(NBATCH):
            runner.step(i)
        merged = gather_audio_to_root(dist, rank, world, mine, [np.stack(a) for a in be.audio], len(allc))
        if rank == 0:
            q.put((merged, runner.bytes_broadcast))
    finally:
        dist.destroy_process_group()


def test_two_rank_raw_broadcast_matches_single_process():
    """only rank 0 ever sees the sample ring; the other rank's clients hear exactly the same audio"""
    import torch
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    allc = _clients()
    be = OracleBackend(torch, _halves(), allc)
    from phantomsdr_amd.distributed import ShardedRunner
    r1 = ShardedRunner(be, None, 0, 1, F)
    for i in range(NBATCH):
        r1.step(i)
    ref = [np.stack(a) for a in be.audio]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_raw_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, nbytes = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert nbytes == (NBATCH * F + 1) * (N // 2) * 8       # every half-frame crosses the wire once
    for a, b in zip(merged, ref):
        assert np.array_equal(a, b)


# ---------------------------------------------------------------------------------------
# band sharding (SURVEY 8e variant ii): rank g receives only the bins its clients can read
# ---------------------------------------------------------------------------------------
HALO = NAUD


def _band_clients():
    """windows (client bins) in both halves of the spectrum, one straddling the band edge (it belongs to
    the band of its left edge and reads the halo), one touching each end of the spectrum"""
    out = [("USB", 0, 0.0, 14), ("AM", 400, 425.0, 450), ("FM", N // 2 - 30, float(N // 2 - 5), N // 2 + 20),
           ("LSB", N // 2 - 14, float(N // 2), N // 2), ("USB", N // 2, N // 2 + 0.5, N // 2 + 14),
           ("AM", 3000, 3025.0, 3050), ("LSB", N - 15, float(N - 1), N - 1)]
    return out


class OracleBandBackend:
    BASE = N // 2 + 1   # client bin c is FFT bin (c + BASE) mod N (src/websocket.cpp:157-160)

    def __init__(self, torch, halves, my_clients, rank, world):
        from oracle import oracle as O
        from phantomsdr_amd.distributed import band_bounds
        self.O, self.torch, self.halves, self.world = O, torch, halves, world
        self.bb = band_bounds
        self.fo = O.FFT(N, False, 3, 0, NAUD)
        self.first, self.bins = band_bounds(rank, N, world, HALO)
        self.band = torch.zeros((2, F, self.bins), dtype=torch.complex64)
        self.specs = None
        self.clients = []
        for mode, l, m, r in my_clients:
            c = O.AudioClient(False, NAUD, 12000, N)
            c.set_audio_demodulation(mode)
            c.set_audio_range(l, m, r)
            self.clients.append(c)
        self.audio = [[] for _ in my_clients]

    def forward(self, i):
        self.specs = []
        for f in range(F):
            g = i * F + f
            self.fo.load(self.halves[g], self.halves[g + 1])
            self.fo.execute()
            self.specs.append(self.fo.output()[:N].copy())

    def pack_bands(self, par):
        out = []
        for g in range(self.world):
            first, bins = self.bb(g, N, self.world, HALO)
            k = ((first + np.arange(bins)) % N + self.BASE) % N
            out.append(self.torch.from_numpy(np.stack([s[k] for s in self.specs])))
        return out

    def band_tensor(self, par):
        return self.band[par]

    def demod_band(self, first_frame_num, par):
        k = ((self.first + np.arange(self.bins)) % N + self.BASE) % N
        for f in range(F):
            full = np.zeros(N + NAUD, np.complex64)       # only the band is known on this rank
            full[k] = self.band[par, f].numpy()
            full[N:] = full[:NAUD]                         # the wrap copy of src/fft.cpp:96-97
            for ci, c in enumerate(self.clients):
                a, _, _, _ = c.send_audio(full, first_frame_num + f, fft=self.fo)
                self.audio[ci].append(a)


def _band_worker(rank, world, port, q, pipelined=True):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from phantomsdr_amd.distributed import BandShardedRunner, assign_clients_by_band, gather_audio_to_root
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        allc = _band_clients()
        mine = assign_clients_by_band([(l, r) for _, l, _, r in allc], N, world, HALO)[rank]
        be = OracleBandBackend(torch, _halves() if rank == 0 else None, [allc[i] for i in mine], rank, world)
        runner = BandShardedRunner(be, dist, rank, world, F, pipelined=pipelined)
        for i in range(NBATCH):
            runner.step(i)
            if pipelined:   # results arrive one step late
                assert all(len(a) == i * F for a in be.audio)
        runner.flush()
        merged = gather_audio_to_root(dist, rank, world, mine, [np.stack(a) for a in be.audio], len(allc))
        if rank == 0:
            q.put((merged, runner.bytes_broadcast, [len(x) for x in
                                                    assign_clients_by_band([(l, r) for _, l, _, r in allc], N, world, HALO)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pipelined", [True, False])
def test_two_rank_band_sharding_matches_single_process(pipelined):
    """each rank sees HALF the spectrum (+ one window of halo) and its clients hear the same audio; with the
    scatter of batch i in flight beside the transform of batch i+1 (pipelined) or in lock step"""
    import torch
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from phantomsdr_amd.distributed import ShardedRunner, band_bounds
    allc = _band_clients()
    be = OracleBackend(torch, _halves(), allc)
    r1 = ShardedRunner(be, None, 0, 1, F)
    for i in range(NBATCH):
        r1.step(i)
    ref = [np.stack(a) for a in be.audio]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_band_worker, args=(r, 2, port, q, pipelined)) for r in range(2)]
    for p in procs:
        p.start()
    merged, nbytes, split = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert split == [4, 3]                                  # both ranks serve clients
    bins = band_bounds(0, N, 2, HALO)[1]
    assert bins == N // 2 + 1 + HALO and nbytes == NBATCH * F * bins * 8   # half a spectrum per link and frame
    for a, b in zip(merged, ref):
        assert np.array_equal(a, b)


def test_band_assignment_rejects_a_window_wider_than_the_halo():
    from phantomsdr_amd.distributed import assign_clients_by_band, band_bounds, band_of
    with pytest.raises(ValueError):
        assign_clients_by_band([(N // 2 - 10, N // 2 + 200)], N, 2, HALO)
    # every left edge has exactly one band, bands tile the spectrum for any G (also non powers of two)
    for G in (1, 2, 3, 7, 8):
        firsts = [band_bounds(g, N, G, HALO)[0] for g in range(G)]
        assert firsts[0] == 0 and firsts == sorted(firsts)
        for l in list(range(0, N, 97)) + [N - 1] + firsts + [f - 1 for f in firsts[1:]]:
            g = band_of(l, N, G)
            first, cnt = band_bounds(g, N, G, HALO)
            assert first <= l < first + cnt - (HALO if G > 1 else 0)   # (G = 1: the band is the spectrum)


def test_banded_regions_contain_the_packed_bands():
    """psdr_set_band_layout's regions (whole columns of 1024 bins, halo rounded up) hold everything psdr_pack_band's
    linear bands hold, for every rank count the banded layout supports: the client assignment does not change"""
    from phantomsdr_amd.distributed import band_bounds, banded_bounds
    R = 1 << 20
    for G in (1, 2, 4, 8, 16):
        for halo in (0, 248, 360, 720, 1024, 1500):
            sizes = set()
            for g in range(G):
                f0, n0 = band_bounds(g, R, G, halo)
                f1, n1 = banded_bounds(g, R, G, halo)
                assert f1 % 1024 == 0 and n1 % 1024 == 0 and f1 == g * (R // G)
                if G > 1:
                    assert f1 <= f0 and f0 + n0 <= f1 + n1 + (1 if halo % 1024 == 0 else 0)
                sizes.add(n1)
            assert len(sizes) == 1  # the scatter wants equal pieces


def test_time_sharding_refuses_the_post_chain():
    from phantomsdr_amd.distributed import TimeShardedRunner

    class _B:
        post_chain = True

    with pytest.raises(ValueError):
        TimeShardedRunner(_B(), 0, 2, 4)


# ---------------------------------------------------------------------------------------
# time sharding: batch g -> rank g mod G with a two-frame warm-up and NO exchange
# ---------------------------------------------------------------------------------------
TF, TSTEPS = 4, 3     # frames per batch, steps per rank (world 2 -> 24 frames in total)


class OracleTimeBackend:
    """TimeShardedRunner back-end on the CPU oracle: fresh (zero-state) clients per run, like
    a GPU that has never seen the frames before its warm-up."""

    def __init__(self, halves, clients):
        from oracle import oracle as O
        self.O, self.halves, self.specs = O, halves, clients
        self.fo = O.FFT(N, False, 3, 0, NAUD)
        self.out = None

    def run(self, first_half, nframes, first_frame_num):
        O = self.O
        cl = []
        for mode, l, m, r in self.specs:
            c = O.AudioClient(False, NAUD, 12000, N)
            c.set_audio_demodulation(mode)
            c.set_audio_range(l, m, r)
            cl.append(c)
        self.out = np.zeros((len(cl), nframes, NAUD // 2), np.float32)
        for f in range(nframes):
            self.fo.load(self.halves[first_half + f], self.halves[first_half + f + 1])
            self.fo.execute()
            for ci, c in enumerate(cl):
                self.out[ci, f], _, _, _ = c.send_audio(self.fo.output(), first_frame_num + f, fft=self.fo)


def _time_halves():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import synth_stream
    nfr = 2 * TSTEPS * TF
    x = synth_stream((nfr + 1) * (N // 2), False, seed=9, fft_size=N).astype(np.complex64)
    return x.reshape(nfr + 1, N // 2)


def _time_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from phantomsdr_amd.distributed import TimeShardedRunner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        be = OracleTimeBackend(_time_halves(), _clients())
        runner = TimeShardedRunner(be, rank, world, TF)
        got = {}
        for s in range(TSTEPS):
            first, skip = runner.step(s)
            got[first] = be.out[:, skip:, :].copy()      # warm-up frames are discarded
        out = [None] * world
        dist.all_gather_object(out, got)                  # result collection only (not data path)
        if rank == 0:
            merged = {}
            for d in out:
                merged.update(d)
            q.put(merged)
    finally:
        dist.destroy_process_group()


def test_time_sharding_matches_sequential_run():
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    halves = _time_halves()
    nfr = 2 * TSTEPS * TF
    seq = OracleTimeBackend(halves, _clients())
    seq.run(0, nfr, 0)                                    # one process, all frames in order
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_time_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(merged) == [g * TF for g in range(2 * TSTEPS)]
    for first, block in merged.items():
        assert block.shape[1] == TF
        # the two warm-up frames rebuild the overlap-add tails and FM's last sample exactly
        assert np.array_equal(block, seq.out[:, first:first + TF, :]), first
