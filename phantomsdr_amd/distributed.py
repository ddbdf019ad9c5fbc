"""Multi-GPU operation: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no multi-GPU path; this is new design.  Ways to shard the path (bench.py --shard; its
default is CLIENT sharding with a spectrum broadcast, the shape BASELINE.json names):

TIME sharding (`--shard time`, `TimeShardedRunner`): the stream itself is the partitioned object.
Frames are independent for the forward FFT and for every client's inverse transform; the
only cross-frame state is the overlap-add tail (second half of the previous frame's
transform) and FM's last sample, and both are functions of the two preceding frames alone.
So batch g (frames [g*F, (g+1)*F)) goes to rank g mod G, which additionally re-runs the two
frames before its batch as warm-up (their outputs are discarded, they only rebuild the
tails): ingest scales with the number of GPUs, each GPU reads its own slice of the sample
ring (its own PCIe link in a deployment), and there is NO data-path collective at all.
Cost: (F+2)/F work per batch.  (The reference's stale cross-mode state - buffers that
survive a mode switch, src/signal.cpp:316-328 - is carried only within a rank.)

CLIENT sharding (bench.py's default, `ShardedRunner`, the shape BASELINE.json's north_star names): after ONE
exchange step the audio clients are independent units, so they shard across ranks:

  rank 0 ("ingest")  owns the raw sample ring, runs the forward FFT + waterfall pyramid
                     and serves the waterfall clients (they only read the int8 pyramid);
  every rank         receives the normalised spectrum batch (F x N complex64, client
                     order, 8*N bytes per frame) through ONE broadcast per batch and
                     demodulates its own clients: client i lives on rank i mod G, its
                     overlap-add state never leaves that GPU.

xGMI is point to point (7 links x ~153 GB/s per GPU), so a 1 -> G-1 broadcast is bound
per link: 8.39 MB per 2^20-point frame / 153 GB/s = 55 us/frame; frames are batched F per
collective to amortise the launch.  With G = 1 there is no collective at all.

BAND sharding (`BandShardedRunner`, SURVEY 8e variant ii): a client only reads the bins of its own
window, so a GPU that serves the clients of ONE frequency band needs only that band: rank g owns the
clients whose window starts in bins [g*R/G, (g+1)*R/G) and receives those bins plus a halo of one
maximal window (a client is assigned by its left edge), R/G + halo bins instead of R: one scatter
per batch, 8.39 MB / G per frame and link.  The per-link ceiling of the ingest rate grows with G
(9.5 GS/s x G for 2^20-point IQ frames) - the client sharding that scales.

The orchestration below is backend-agnostic (the compute back-end is injected), so the
sharding and the exchange are covered by world_size-2 gloo tests on CPU
(tests/test_distributed_cpu.py) with the oracle as compute stand-in, and by the HIP
back-end on the GPUs.
"""
import contextlib

import numpy as np


def assign_clients(nclients, world):
    """client i -> rank i mod G (SURVEY 8e "gpu = hash(client_id) mod G")."""
    return [list(range(r, nclients, world)) for r in range(world)]


class ShardedRunner:
    """Drives one batch per step() on every rank.

    backend must provide:
      forward(step_index)      rank 0 only: fill spectrum_tensor() with F fresh spectra
      spectrum_tensor()        torch tensor (same shape/dtype on every rank) to broadcast
      demod(first_frame_num)   demodulate this rank's clients from spectrum_tensor()
    """

    def __init__(self, backend, dist, rank, world, frames_per_step, root=0):
        self.backend, self.dist = backend, dist
        self.rank, self.world, self.F, self.root = rank, world, frames_per_step, root
        self.frame_num = 0
        self.bytes_broadcast = 0

    def step(self, i):
        # the forward kernels, the collective and the demodulation are ordered on ONE stream
        # (the back-end's; a CPU back-end has none)
        ctx = getattr(self.backend, "stream_context", None)
        with (ctx() if ctx else contextlib.nullcontext()):
            if self.rank == self.root:
                self.backend.forward(i)
            if self.world > 1:
                t = self.backend.spectrum_tensor()
                self.dist.broadcast(t, src=self.root)
                self.bytes_broadcast += t.numel() * t.element_size()
            self.backend.demod(self.frame_num)
        self.frame_num += self.F


class PipelinedShardedRunner:
    """ShardedRunner with the broadcast of batch i running beside the root's transform of batch i+1 and
    everybody's demodulation of batch i-1 (the same one-step software pipeline as BandShardedRunner, with
    the whole spectrum as the one "band"): the step rate becomes max(compute, link time) instead of their
    sum.  The root copies each finished spectrum batch out of the transform's own buffer into one of two
    send buffers (that copy is what lets the next transform start); results arrive one step late, flush()
    delivers the last batch.  backend: forward(i) (root), stage(par) (root: fill spectrum_tensor(par)),
    spectrum_tensor(par), demod(first_frame_num, par)."""

    def __init__(self, backend, dist, rank, world, frames_per_step, root=0):
        self.backend, self.dist = backend, dist
        self.rank, self.world, self.F, self.root = rank, world, frames_per_step, root
        self.frame_num = 0
        self.bytes_broadcast = 0
        self._n = 0
        self._pending = None

    def _ctx(self):
        ctx = getattr(self.backend, "stream_context", None)
        return ctx() if ctx else contextlib.nullcontext()

    def _drain(self):
        work, par, first = self._pending
        self._pending = None
        if work is not None:
            work.wait()
        self.backend.demod(first, par)

    def step(self, i):
        par = self._n & 1
        with self._ctx():
            if self.rank == self.root:
                self.backend.forward(i)
                self.backend.stage(par)
            work = None
            if self.world > 1:
                t = self.backend.spectrum_tensor(par)
                work = self.dist.broadcast(t, src=self.root, async_op=True)
                self.bytes_broadcast += t.numel() * t.element_size()
            if self._pending is not None:
                self._drain()
            self._pending = (work, par, self.frame_num)
        self.frame_num += self.F
        self._n += 1

    def flush(self):
        if self._pending is not None:
            with self._ctx():
                self._drain()


class RawShardedRunner:
    """SURVEY 8e variant (i): clients sharded as in ShardedRunner, but what crosses xGMI is the RAW
    new half-frames (cs16: 2.1 MB per 2^20-point frame, 4x fewer bytes than the 8.39 MB spectrum)
    and every rank runs the forward FFT itself.  The broadcast's per-link ceiling moves from
    ~9.5 to ~38 GS/s of ingest; the price is G redundant forward transforms, i.e. per-GPU compute
    stays that of a single GPU and only the clients scale.

    backend must provide:
      raw_tensor()                 torch tensor of the batch's F+1 raw half-frames (same shape on
                                   every rank); row 0 is the last half of the previous batch
      load_raw(step_index)         root only: fill rows 1..F (and row 0 at step 0) from the ring
      roll()                       every rank, before the next step: row F -> row 0
      forward_local()              forward FFT (+ pyramid) of the F frames in raw_tensor()
      demod(first_frame_num)       demodulate this rank's clients
    """

    def __init__(self, backend, dist, rank, world, frames_per_step, root=0):
        self.backend, self.dist = backend, dist
        self.rank, self.world, self.F, self.root = rank, world, frames_per_step, root
        self.frame_num = 0
        self.bytes_broadcast = 0
        self.first = True

    def step(self, i):
        ctx = getattr(self.backend, "stream_context", None)
        with (ctx() if ctx else contextlib.nullcontext()):
            if not self.first:
                self.backend.roll()
            if self.rank == self.root:
                self.backend.load_raw(i)
            if self.world > 1:
                t = self.backend.raw_tensor()
                t = t if self.first else t[1:]   # row 0 is already everywhere after the first step
                # the collective moves BYTES: RCCL/NCCL carry no 16-bit integer type (torch refuses int16 on the
                # nccl backend, gloo answers "Invalid scalar type") - found by the two-process run of bench.py on
                # one GPU (PSDR_BENCH_ONE_DEVICE), round 3
                t = getattr(self.backend, "wire_view", lambda x: x)(t)
                self.dist.broadcast(t, src=self.root)
                self.bytes_broadcast += t.numel() * t.element_size()
            self.backend.forward_local()
            self.backend.demod(self.frame_num)
        self.first = False
        self.frame_num += self.F


def band_of(l, R, world):
    """rank of the band a window starting at bin l belongs to"""
    return min(int(l) * world // R, world - 1)


def band_bounds(g, R, world, halo):
    """(first bin, bins) of rank g's band: [g*R/G, (g+1)*R/G) plus `halo` bins, the same count on every
    rank (the scatter wants equal pieces; the last band's halo wraps and is never read)"""
    first = (g * R + world - 1) // world       # smallest l with band_of(l) == g
    return first, min((R + world - 1) // world + 1 + halo, R)


def assign_clients_by_band(windows, R, world, halo):
    """windows: [(l, r)] in client bins -> list of client indices per rank; raises if a window does not
    fit its band (it is wider than the halo)"""
    out = [[] for _ in range(world)]
    for i, (l, r) in enumerate(windows):
        g = band_of(l, R, world)
        first, cnt = band_bounds(g, R, world, halo)
        if not (first <= l and r <= first + cnt):
            raise ValueError(f"client {i}: window [{l}, {r}) does not fit band {g} = [{first}, {first + cnt})")
        out[g].append(i)
    return out


class BandShardedRunner:
    """SURVEY 8e variant (ii).  backend must provide (par = 0/1: which of two buffer sets)
      forward(step_index)             root only: F fresh spectra
      pack_bands(par)                 root only: list of G tensors [F, bins] (band g for rank g), contiguous
      band_tensor(par)                this rank's receive tensor [F, bins]
      demod_band(first_frame_num, par)  demodulate this rank's clients from band_tensor(par)

    pipelined (default): the scatter of batch i is asynchronous and is only waited for one step later,
    right before batch i is demodulated - so it runs beside the root's forward transform + pack of
    batch i+1 and beside everybody's demodulation of batch i-1 (send and receive buffers alternate).
    The step rate is max(compute, link time) instead of their sum; results arrive one step late and
    flush() delivers the last batch.  The order of operations per client is unchanged, so the audio is
    bit-identical to the unpipelined and to the unsharded run.
    """

    def __init__(self, backend, dist, rank, world, frames_per_step, root=0, pipelined=True):
        self.backend, self.dist = backend, dist
        self.rank, self.world, self.F, self.root = rank, world, frames_per_step, root
        self.pipelined = pipelined
        self.frame_num = 0
        self.bytes_broadcast = 0   # bytes that left the root per link (one band per peer)
        self._n = 0
        self._pending = None

    def _ctx(self):
        ctx = getattr(self.backend, "stream_context", None)
        return ctx() if ctx else contextlib.nullcontext()

    def _drain(self):
        work, par, first = self._pending
        self._pending = None
        if work is not None:
            work.wait()          # NCCL: the back-end's stream waits; gloo: the host does
        self.backend.demod_band(first, par)

    def step(self, i):
        par = self._n & 1
        with self._ctx():
            bands = None
            if self.rank == self.root:
                self.backend.forward(i)
                bands = self.backend.pack_bands(par)
            t = self.backend.band_tensor(par)
            work = None
            if self.world > 1:
                work = self.dist.scatter(t, scatter_list=bands if self.rank == self.root else None, src=self.root,
                                         async_op=self.pipelined)
                self.bytes_broadcast += t.numel() * t.element_size()
            elif getattr(self.backend, "adopt", None) and self.backend.adopt(bands[0], par):
                pass             # one rank, nothing to ship: the region is demodulated where pass 2 wrote it
            else:
                t.copy_(bands[0])
            if self.pipelined:
                if self._pending is not None:
                    self._drain()
                self._pending = (work, par, self.frame_num)
            else:
                self.backend.demod_band(self.frame_num, par)
        self.frame_num += self.F
        self._n += 1

    def flush(self):
        """demodulate the batch still in flight (call after the last step)"""
        if self._pending is not None:
            with self._ctx():
                self._drain()


class TimeShardedRunner:
    """Batch g of the stream -> rank g mod G, with a two-frame warm-up instead of any
    exchange.  backend must provide
      run(first_half, nframes, first_frame_num)   forward + demodulate `nframes` frames that
                                                  start at half-frame `first_half`
      collect(skip)                               per-client audio of the last run without
                                                  the first `skip` (warm-up) frames
    """
    WARMUP = 2

    def __init__(self, backend, rank, world, frames_per_step):
        # The two-frame warm-up rebuilds the overlap-add tails and FM's last sample exactly, nothing
        # more: the post-demodulation chain (DC blocker sums, 200 ms AGC look-ahead and gain) has
        # seconds of memory and the NaN guard can drop warm-up frames - time sharding would
        # silently change the PCM there, so it is refused.
        if getattr(backend, "post_chain", False):
            raise ValueError("time sharding is exact only up to the float audio: disable the post-demodulation "
                             "chain (psdr_set_post_chain) or shard the clients instead")
        self.backend, self.rank, self.world, self.F = backend, rank, world, frames_per_step
        self.step_index = 0

    def batch_of(self, step):
        return step * self.world + self.rank

    def step(self, step=None):
        """runs this rank's batch of global step `step`; returns (first_frame, skip)"""
        s = self.step_index if step is None else step
        g = self.batch_of(s)
        first = g * self.F
        skip = min(self.WARMUP, first)  # the very first batch of the stream has no past
        self.backend.run(first - skip, self.F + skip, first - skip)
        self.step_index = s + 1
        return first, skip


class HipTimeBackend:
    """TimeShardedRunner back-end on the HIP library (ring = device pointer to raw halves;
    the ring is cycled modulo `ring_batches` batches for the benchmark)."""

    def __init__(self, ctx, ring_ptr, ring_halves, max_frames):
        self.ctx, self.ring_ptr, self.ring_halves = ctx, ring_ptr, ring_halves
        self.hb = ctx.half_frame_bytes()
        self.max_frames = max_frames
        self.last = (0, 0)
        self.post_chain = bool(getattr(ctx, "post_chain_on", False))

    def run(self, first_half, nframes, first_frame_num):
        span = nframes + 1
        off = first_half % max(1, self.ring_halves - span)  # stay inside the synthetic ring
        self.ctx.process_batch(self.ring_ptr, nframes, offset_bytes=off * self.hb)
        self.ctx.demod_batch(first_frame_num)
        self.last = (first_frame_num, nframes)


class _CudaArray:
    """minimal __cuda_array_interface__ carrier so torch can alias library-owned HBM"""

    def __init__(self, ptr, nelem, typestr):
        self.__cuda_array_interface__ = {"shape": (int(nelem),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def alias_device_f32(torch, ptr, nfloats, device):
    """torch float32 tensor aliasing `nfloats` floats of device memory at `ptr`"""
    return torch.as_tensor(_CudaArray(ptr, nfloats, "<f4"), device=device)


class HipBackend:
    """ShardedRunner back-end on the HIP library: every rank owns a Context; the
    broadcast lands directly in the context's own spectrum buffer."""

    def __init__(self, torch, ctx, device, ring_ptr, nbatches, frames_per_step):
        import ctypes as C
        self.torch, self.ctx, self.F = torch, ctx, frames_per_step
        self.ring_ptr, self.nbatches = ring_ptr, nbatches
        self.hb = ctx.half_frame_bytes()
        from ._lib import check
        # ONE stream for the kernels and the collective's ordering.  It must be a stream of its own:
        # torch's default stream has handle 0, which psdr_set_stream() reads as "restore the
        # context's own streams" - the context would then alternate between its two result sets
        # while the broadcast keeps sending set 0 (found by tests/test_gpu_fullsize.py).
        self.stream = torch.cuda.Stream(device=device)
        assert self.stream.cuda_stream != 0
        check(ctx.lib.psdr_set_stream(ctx.h, C.c_void_p(self.stream.cuda_stream)))
        # (after psdr_set_stream: on a caller's stream the context keeps to ONE result set)
        p, nb = C.c_void_p(), C.c_size_t()
        check(ctx.lib.psdr_spectrum_device_ptr(ctx.h, 0, C.byref(p), C.byref(nb)))
        self.spec_ptr = p.value
        self.stride_bins = ctx.N if not ctx.is_real else ctx.N // 2 + 2
        self.spec = alias_device_f32(torch, self.spec_ptr, self.F * self.stride_bins * 2, device)

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def synchronize(self):
        self.stream.synchronize()

    def forward(self, i):
        b = i % self.nbatches
        self.ctx.process_batch(self.ring_ptr, self.F, offset_bytes=b * self.F * self.hb)

    def spectrum_tensor(self):
        return self.spec

    def demod(self, first_frame_num):
        import ctypes as C
        from ._lib import check
        check(self.ctx.lib.psdr_demod_batch_from(self.ctx.h, C.c_void_p(self.spec_ptr),
                                                 self.stride_bins, self.F, first_frame_num))
        self.ctx.last_nframes = self.ctx.last_demod_frames = self.F


def banded_bounds(g, R, world, halo, column=1024):
    """(first bin, bins) of rank g's band REGION in a banded spectrum (psdr_set_band_layout): whole columns of
    `column` bins, the halo rounded up to columns.  Contains band_bounds(g, ...) for every g."""
    per = R // world
    hcols = -(-halo // column)
    if hcols * column > per:  # (psdr_set_band_layout refuses it too: a halo must be the NEXT band's own columns)
        raise ValueError(f"halo of {halo} bins exceeds a band of {per} bins ({world} bands)")
    return g * per, per + hcols * column


class HipBandBackend:
    """BandShardedRunner back-end on the HIP library.

    banded (default where the library supports it: 2^20-point IQ frames, world a power of two <= 16): the root's
    second FFT pass writes the spectrum as one contiguous region per band (psdr_set_band_layout) and the regions ARE
    the send buffers - no pack pass, no second copy of the spectrum on the root; every rank demodulates from its region
    (psdr_demod_batch_from_band_region).  Two result sets alternate inside the library, so the region of batch b is
    stable while batch b+1 is transformed and the (asynchronous) scatter of b is in flight.

    Otherwise the root packs the G bands out of the device layout (psdr_pack_band, one small kernel per band) into
    one send buffer and every rank demodulates from its linear band buffer (psdr_demod_batch_from_band)."""

    def __init__(self, torch, ctx, device, ring_ptr, nbatches, frames_per_step, rank, world, halo, root=0, banded=None):
        import ctypes as C
        from ._lib import check
        self.torch, self.ctx, self.F = torch, ctx, frames_per_step
        self.ring_ptr, self.nbatches, self.rank, self.world = ring_ptr, nbatches, rank, world
        self.hb = ctx.half_frame_bytes()
        self.R = ctx.N // 2 if ctx.is_real else ctx.N
        self.halo = halo
        self.device, self.root = device, root
        self.stream = torch.cuda.Stream(device=device)
        assert self.stream.cuda_stream != 0
        check(ctx.lib.psdr_set_stream(ctx.h, C.c_void_p(self.stream.cuda_stream)))
        can = (not ctx.is_real) and ctx.N in (1 << 20, 1 << 21) and world <= 16 and world & (world - 1) == 0
        if banded is None:
            banded = can
        if banded and not can:
            raise ValueError("banded band sharding: 2^20- or 2^21-point IQ frames and a power-of-two world <= 16")
        self.banded = banded
        if banded:
            self.column = ctx.N >> 10  # bins per column of the (c1, c2) grid: M1
            self.first, self.bins = banded_bounds(rank, self.R, world, halo, self.column)
            if rank == root:  # (a receiver's context keeps its own layout: it never transforms)
                check(ctx.lib.psdr_set_band_layout(ctx.h, world, halo))
            self.send = None
        else:
            self.first, self.bins = band_bounds(rank, self.R, world, halo)
            self.send = (torch.empty((2, world, frames_per_step, self.bins), dtype=torch.complex64, device=device)
                         if rank == root else None)
        self.band = torch.empty((2, frames_per_step, self.bins), dtype=torch.complex64, device=device)
        self._adopted = {}

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def synchronize(self):
        self.stream.synchronize()

    def forward(self, i):
        b = i % self.nbatches
        self.ctx.process_batch(self.ring_ptr, self.F, offset_bytes=b * self.F * self.hb)

    def _regions(self):
        """the G band regions of the batch just transformed, as tensors aliasing the library's buffer"""
        import ctypes as C
        from ._lib import check
        out = []
        for g in range(self.world):
            p, fs, fb, nb = C.c_void_p(), C.c_size_t(), C.c_uint32(), C.c_uint32()
            check(self.ctx.lib.psdr_band_region(self.ctx.h, g, C.byref(p), C.byref(fs), C.byref(fb), C.byref(nb)))
            assert (fb.value, nb.value) == banded_bounds(g, self.R, self.world, self.halo, self.column) and fs.value == nb.value
            t = alias_device_f32(self.torch, p.value, self.F * nb.value * 2, self.device)
            out.append(self.torch.view_as_complex(t.view(self.F, nb.value, 2)))
        return out

    def pack_bands(self, par=0):
        import ctypes as C
        from ._lib import check
        if self.banded:
            return self._regions()
        for g in range(self.world):
            first, bins = band_bounds(g, self.R, self.world, self.halo)
            check(self.ctx.lib.psdr_pack_band(self.ctx.h, self.F, first, bins, C.c_void_p(self.send[par, g].data_ptr()), bins))
        return [self.send[par, g] for g in range(self.world)]

    def band_tensor(self, par=0):
        return self.band[par]

    def adopt(self, region, par=0):
        """world = 1 (bench.py --force-sharded on one GPU): demodulate band 0 in place instead of copying it into the
        receive buffer.  PSDR_BAND_STANDIN=1 keeps the copy as a stand-in for the link transfer (it over-states what a
        scatter costs the root: the peers only READ the regions)."""
        import os
        if not self.banded or os.environ.get("PSDR_BAND_STANDIN") == "1":
            return False
        self._adopted[par] = region
        return True

    def demod_band(self, first_frame_num, par=0):
        import ctypes as C
        from ._lib import check
        fn = self.ctx.lib.psdr_demod_batch_from_band_region if self.banded else self.ctx.lib.psdr_demod_batch_from_band
        src = self._adopted.pop(par, None)
        if src is None:
            src = self.band[par]
        check(fn(self.ctx.h, C.c_void_p(src.data_ptr()), self.bins, self.first, self.bins, self.F, first_frame_num))
        self.ctx.last_nframes = self.ctx.last_demod_frames = self.F


class HipPipelinedBackend:
    """PipelinedShardedRunner back-end on the HIP library: the staged copy is psdr_pack_band over the whole
    spectrum (linear, client order), the receivers demodulate with psdr_demod_batch_from_band."""

    def __init__(self, torch, ctx, device, ring_ptr, nbatches, frames_per_step, is_root):
        import ctypes as C
        from ._lib import check
        self.torch, self.ctx, self.F = torch, ctx, frames_per_step
        self.ring_ptr, self.nbatches = ring_ptr, nbatches
        self.hb = ctx.half_frame_bytes()
        self.R = ctx.N // 2 if ctx.is_real else ctx.N
        self.stream = torch.cuda.Stream(device=device)
        assert self.stream.cuda_stream != 0
        check(ctx.lib.psdr_set_stream(ctx.h, C.c_void_p(self.stream.cuda_stream)))
        self.buf = torch.empty((2, frames_per_step, self.R), dtype=torch.complex64, device=device)

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def synchronize(self):
        self.stream.synchronize()

    def forward(self, i):
        b = i % self.nbatches
        self.ctx.process_batch(self.ring_ptr, self.F, offset_bytes=b * self.F * self.hb)

    def stage(self, par):
        import ctypes as C
        from ._lib import check
        check(self.ctx.lib.psdr_pack_band(self.ctx.h, self.F, 0, self.R, C.c_void_p(self.buf[par].data_ptr()), self.R))

    def spectrum_tensor(self, par):
        return self.buf[par]

    def demod(self, first_frame_num, par):
        import ctypes as C
        from ._lib import check
        check(self.ctx.lib.psdr_demod_batch_from_band(self.ctx.h, C.c_void_p(self.buf[par].data_ptr()), self.R, 0, self.R,
                                                      self.F, first_frame_num))
        self.ctx.last_nframes = self.ctx.last_demod_frames = self.F


class HipRawBackend:
    """RawShardedRunner back-end on the HIP library: every rank owns a Context and a device buffer of
    F+1 raw half-frames; the broadcast lands in that buffer, psdr_process_batch reads it."""

    def __init__(self, torch, ctx, device, ring, nbatches, frames_per_step):
        import ctypes as C
        from ._lib import check
        self.torch, self.ctx, self.F = torch, ctx, frames_per_step
        self.ring, self.nbatches = ring, nbatches  # ring: torch int16 tensor [halves][samples] on the root, else None
        self.hb = ctx.half_frame_bytes()
        self.stream = torch.cuda.Stream(device=device)
        check(ctx.lib.psdr_set_stream(ctx.h, C.c_void_p(self.stream.cuda_stream)))
        self.raw = torch.zeros((self.F + 1, self.hb // 2), dtype=torch.int16, device=device)

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def raw_tensor(self):
        return self.raw

    def wire_view(self, t):
        """what a collective is given: the same memory as unsigned bytes"""
        return t.view(self.torch.uint8)

    def load_raw(self, i):
        b = i % self.nbatches
        src = self.ring[b * self.F: b * self.F + self.F + 1]
        if i == 0:
            self.raw.copy_(src)
        else:
            self.raw[1:].copy_(src[1:])

    def roll(self):
        self.raw[0].copy_(self.raw[self.F])

    def forward_local(self):
        self.ctx.process_batch(self.raw.data_ptr(), self.F)

    def demod(self, first_frame_num):
        self.ctx.demod_batch(first_frame_num)


def gather_audio_to_root(dist, rank, world, local_ids, local_audio, nclients, root=0):
    """collects per-client audio blocks [F][n/2] on the root in global client order
    (results normally go to the host directly from each GPU; this is for tests/tools)."""
    payload = {cid: np.asarray(a) for cid, a in zip(local_ids, local_audio)}
    out = [None] * world
    dist.all_gather_object(out, payload)
    if rank != root:
        return None
    merged = {}
    for d in out:
        merged.update(d)
    return [merged[i] for i in range(nclients)]
