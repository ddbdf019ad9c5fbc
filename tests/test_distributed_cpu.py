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
