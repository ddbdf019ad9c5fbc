"""Oracle parity at BASELINE.json's FULL sizes with the bench's own client sets.

Every BASELINE configuration's single-GPU shape runs here through the C-ABI exactly as
bench.py drives it - raw s16 ring in HBM -> psdr_process_batch -> psdr_demod_batch ->
psdr_waterfall_batch - for a few frames, and EVERYTHING is compared with the CPU oracle:
spectrum (1e-4 of the peak, 1e-5 relative L2), int8 pyramid (bit-exact against the reference
quantiser on the GPU's own spectrum, >= 99.9 % / +-1 against the oracle's), every client's
audio and pwr (1e-4), and the gathered waterfall rows.

  cfg2  35 MSPS IQ cs16, 2^20-pt C2C, 16 SSB + 4 waterfall          (src/fft.cpp:47-105)
  cfg3  70 MSPS real s16, 2^21-pt R2C, 64 mixed AM/FM/SSB           (src/signal.cpp:102-275)
  cfg4  cfg2's spectrum with 32 mixed clients through psdr_demod_batch_from + psdr_set_stream,
        i.e. the receiving GPU's share of the client-sharded run    (src/websocket.cpp:156-185)
  cfg5  70 MSPS real s16, 2^22-pt R2C, n = 720, 128 clients + 8 zoomed waterfalls
                                                                     (src/websocket.cpp:207-236)
  c256  cfg2 with 256 mixed audio clients on one GPU (the north-star target's own wording)
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from helpers import check_fm, pwr_tolerance, quantize_raw, rel_err, rel_l2, synth_stream
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

SPEC_TOL, SPEC_L2, AUDIO_TOL = 1e-4, 1e-5, 1e-4


def _bench():
    import bench
    return bench


def _oracle_clients(clients, is_real, n, R):
    out = []
    for mode, l, m, r in clients:
        o = O.AudioClient(is_real, n, 12000, R)
        o.set_audio_demodulation(mode)
        o.set_audio_range(l, m, r)
        out.append(o)
    return out


def _check_audio(tag, mode, a_g, p_g, nan_g, a_o, p_o, dropped, oc=None):
    """oc: the oracle client right after its send_audio (FM: its baseband conditions the bound)"""
    assert not dropped and nan_g == 0, tag
    assert abs(p_g - p_o) <= pwr_tolerance(p_o, oc.fwd_scale if oc is not None else 0.0), f"{tag}: pwr {p_g} vs {p_o}"
    if mode == O.FM:
        # 1e-4 rad where both discriminator inputs are at least 5 % of the peak, scaled by the
        # conditioning below that (helpers.fm_tolerance)
        check_fm(a_g, a_o, oc.baseband()[: oc.n // 2], oc.bb_prev, tag, fwd_scale=max(oc.fwd_scale, oc.fwd_scale_prev))
    else:
        assert rel_l2(a_g, a_o) < AUDIO_TOL, f"{tag}: rel L2 {rel_l2(a_g, a_o):.2e}"
        assert np.abs(a_g - a_o).max() <= 2e-4 * max(np.abs(a_o).max(), 1e-30), tag


def _check_pyramid(q_gpu, spec_gpu_k, q_orc, N, is_real, levels, tag):
    q_self = O.pyramid_from_spectrum(spec_gpu_k, N, is_real, levels)
    assert np.array_equal(q_gpu, q_self), (
        f"{tag}: int8 pyramid differs from the reference quantiser applied to the GPU's own spectrum: "
        f"{(q_gpu != q_self).sum()} of {q_gpu.size}")
    d = np.abs(q_gpu.astype(np.int16) - q_orc.astype(np.int16))
    assert d.max() <= 1, f"{tag}: pyramid differs from oracle by {d.max()} LSB"
    assert (d != 0).mean() <= 1e-3, f"{tag}: pyramid mismatch rate {(d != 0).mean():.2e}"


def run_workload(wl, clients_fn=None, splits=(3, 2), seed=77):
    """bench.py's workload `wl` for sum(splits) frames in batches of `splits`, all against the oracle."""
    from phantomsdr_amd import SpectrumEngine
    B = _bench()
    N, is_real, fmt = wl["fft_size"], wl["is_real"], wl["fmt"]
    nframes, F = sum(splits), max(splits)
    eng = SpectrumEngine(wl["sps"], N, is_real, input_format=fmt, max_batch=F,
                         max_clients=max(wl["audio"], 1), max_waterfall_clients=max(wl["waterfall"], 1))
    try:
        p = eng.params
        R, n, levels, skip = p["fft_result_size"], p["audio_fft_size"], p["downsample_levels"], p["skip_num"]
        clients = (clients_fn or B.make_clients)(wl, p, seed=0x5D5D0002)
        waterfalls = B.make_waterfalls(wl, p, seed=0x5D5D0002)
        x = synth_stream((nframes + 1) * (N // 2), is_real, seed=seed, fft_size=N)
        raw = quantize_raw(x, fmt, is_real)
        del x
        eng.upload_ring(raw)
        conv = O.convert(raw, fmt)
        halves = (conv if is_real else conv.view(np.complex64)).reshape(nframes + 1, N // 2)
        gcl = [eng.add_audio_client(l, m, r, mode) for mode, l, m, r in clients]
        gwf = [eng.add_waterfall_client(lv, l, r) for lv, l, r in waterfalls]
        ocl = _oracle_clients(clients, is_real, n, R)
        fo = O.FFT(N, is_real, levels, 0, n)
        frame = 0
        for nf in splits:
            first = eng.frame_num
            eng.step(frame, nf)
            got = [g.read_audio(nf) for g in gcl]
            wrows = [w.read_waterfall()[0] for w in gwf]
            si = 0
            for f in range(nf):
                fo.load(halves[frame], halves[frame + 1])
                fo.execute()
                spec_o = fo.output().copy()
                tag = f"frame {frame}"
                Xg = eng.ctx.read_spectrum(f)
                nb = N // 2 if is_real else N
                assert rel_err(Xg[:nb], spec_o[:nb]) < SPEC_TOL, tag
                assert rel_l2(Xg[:nb], spec_o[:nb]) < SPEC_L2, tag
                qg = eng.ctx.read_quantized(f)
                _check_pyramid(qg, Xg, fo.quantized().copy(), N, is_real, levels, tag)
                if (first + f) % skip == 0:
                    for wi, (lv, l, r) in enumerate(waterfalls):
                        row_g = wrows[wi][si]
                        # bytes of q_level[l..r) (src/waterfall.cpp:44-51): the GPU's own pyramid exactly,
                        # the oracle's within the quantiser tolerance
                        assert np.array_equal(row_g, eng.ctx.quantized_level(qg, lv)[l:r]), f"{tag} waterfall {wi}"
                        d = np.abs(row_g.astype(np.int16) - fo.quantized_level(lv)[l:r].astype(np.int16))
                        assert d.max() <= 1 and (d != 0).mean() <= 5e-3, f"{tag} waterfall {wi} vs oracle"
                    si += 1
                for ci, o in enumerate(ocl):
                    a_o, p_o, _, dropped = o.send_audio(spec_o, first + f, fft=fo)
                    _check_audio(f"{tag} client {ci} {clients[ci]}", o.mode, got[ci][0][f], got[ci][1][f],
                                 got[ci][2][f], a_o, p_o, dropped, o)
                frame += 1
            for wi in range(len(waterfalls)):
                assert wrows[wi].shape[0] == si, "number of sent waterfall rows"
    finally:
        eng.close()


def test_cfg2_fullsize_vs_oracle():
    """configs[1]: 2^20-pt IQ cs16, the bench's 16 SSB clients + 4 waterfall clients (L = 11)."""
    run_workload(_bench().WORKLOADS["cfg2"], splits=(3, 2))


def test_cfg3_fullsize_vs_oracle():
    """configs[2]: 2^21-pt real s16 (R2C), the bench's 64 mixed AM/FM/SSB clients."""
    run_workload(_bench().WORKLOADS["cfg3"], splits=(2, 2))


def test_cfg5_share_fullsize_vs_oracle():
    """configs[4], one GPU's share: 2^22-pt real s16, n = 720, 128 clients + 8 zoomed waterfalls."""
    run_workload(_bench().WORKLOADS["cfg5"], splits=(2, 1))


def test_clients256_fullsize_vs_oracle():
    """the target's own workload: 2^20-pt IQ with 256 concurrent mixed audio clients on ONE GPU."""
    B = _bench()
    wl = dict(B.WORKLOADS["cfg2"], audio=256, modes=("USB", "LSB", "AM", "FM"))
    run_workload(wl, splits=(2, 1))


def test_cfg4_share_demod_from_matches_unsharded_and_oracle():
    """configs[3], the receiving GPU's side of the client-sharded run, on ONE GPU.  Rank 3's 32 clients
    (of 256 over 8 ranks) are demodulated three ways from the same raw stream:
      A  psdr_process_batch + psdr_demod_batch                      (the unsharded path)
      B  phantomsdr_amd.distributed.HipBackend + ShardedRunner, world = 1: psdr_set_stream(torch's
         stream) + psdr_demod_batch_from on the context's own spectrum buffer (bench --force-sharded)
      C  a context that never runs an FFT: A's spectrum batch is copied device-to-device into a
         foreign buffer with a padded frame stride (the stand-in for the RCCL broadcast's receive
         buffer) and demodulated with psdr_demod_batch_from on a caller-owned stream.
      D  HipRawBackend + RawShardedRunner, world = 1: the raw half-frames go through the broadcast
         buffer (row F carried over to row 0 between steps) and the FFT runs from there.
    B, C and D must equal A bit for bit; A is checked against the oracle."""
    import torch
    from phantomsdr_amd import SpectrumEngine
    from phantomsdr_amd._lib import check
    from phantomsdr_amd.distributed import (HipBackend, HipRawBackend, RawShardedRunner, ShardedRunner,
                                            alias_device_f32, assign_clients)
    B = _bench()
    wl = B.WORKLOADS["cfg4"]
    N, F, world, nb_ = wl["fft_size"], 3, 8, 2
    dev = torch.device("cuda", 0)
    x = synth_stream((nb_ * F + 1) * (N // 2), False, seed=91, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    del x
    ring = torch.from_numpy(raw.view(np.int16)).to(dev)
    torch.cuda.synchronize()
    mk = lambda: SpectrumEngine(wl["sps"], N, False, input_format="s16", max_batch=F, max_clients=wl["audio"],
                                max_waterfall_clients=1)
    engA, engB, engC, engD = mk(), mk(), mk(), mk()
    try:
        p = engA.params
        n, levels, R = p["audio_fft_size"], p["downsample_levels"], p["fft_result_size"]
        all_clients = B.make_clients(wl, p, seed=0x5D5D0004, count=wl["audio"] * world)
        mine = [all_clients[c] for c in assign_clients(len(all_clients), world)[3]]
        assert len(mine) == wl["audio"]
        gA, gB, gC, gD = ([e.add_audio_client(l, m, r, mode) for mode, l, m, r in mine] for e in (engA, engB, engC, engD))
        hb = engA.ctx.half_frame_bytes()
        backend = HipBackend(torch, engB.ctx, dev, ring.data_ptr(), nb_, F)
        runner = ShardedRunner(backend, None, 0, 1, F)
        raw_runner = RawShardedRunner(HipRawBackend(torch, engD.ctx, dev, ring.view(nb_ * F + 1, -1), nb_, F), None, 0, 1, F)
        stride = N + 64
        foreign = torch.zeros(F * stride * 2, dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        check(engC.ctx.lib.psdr_set_stream(engC.ctx.h, C.c_void_p(side.cuda_stream)))
        ocl = _oracle_clients(mine, False, n, R)
        fo = O.FFT(N, False, levels, 0, n)
        conv = O.convert(raw, "s16").view(np.complex64).reshape(nb_ * F + 1, N // 2)
        for b in range(nb_):
            engA.ctx.process_batch(ring.data_ptr(), F, offset_bytes=b * F * hb)
            engA.ctx.demod_batch(b * F)
            engA.ctx.synchronize()
            runner.step(b)
            raw_runner.step(b)
            src, nbins = C.c_void_p(), C.c_size_t()
            check(engA.ctx.lib.psdr_spectrum_device_ptr(engA.ctx.h, 0, C.byref(src), C.byref(nbins)))
            t_src = alias_device_f32(torch, src.value, F * N * 2, dev).view(F, N * 2)
            with torch.cuda.stream(side):
                foreign.view(F, stride * 2)[:, : N * 2].copy_(t_src)
                check(engC.ctx.lib.psdr_demod_batch_from(engC.ctx.h, C.c_void_p(foreign.data_ptr()), stride, F, b * F))
            torch.cuda.synchronize()
            got = [g.read_audio(F) for g in gA]
            for ci in range(len(mine)):
                for name, gx in (("HipBackend", gB), ("foreign buffer", gC), ("HipRawBackend", gD)):
                    a2, p2, n2 = gx[ci].read_audio(F)
                    assert np.array_equal(got[ci][0].view(np.uint32), a2.view(np.uint32)), f"{name}: batch {b} client {ci} audio"
                    assert np.array_equal(got[ci][1].view(np.uint32), p2.view(np.uint32)), f"{name}: batch {b} client {ci} pwr"
                    assert np.array_equal(got[ci][2], n2)
            for f in range(F):
                fr = b * F + f
                fo.load(conv[fr], conv[fr + 1])
                fo.execute()
                spec_o = fo.output().copy()
                for ci, o in enumerate(ocl):
                    a_o, p_o, _, dropped = o.send_audio(spec_o, fr, fft=fo)
                    _check_audio(f"frame {fr} client {ci} {mine[ci]}", o.mode, got[ci][0][f], got[ci][1][f],
                                 got[ci][2][f], a_o, p_o, dropped, o)
    finally:
        for e in (engB, engC, engD):
            check(e.ctx.lib.psdr_set_stream(e.ctx.h, None))
        for e in (engA, engB, engC, engD):
            e.close()


@pytest.mark.parametrize("wl_name,world,g,banded", [("cfg4", 8, 5, True), ("cfg4", 8, 7, True), ("cfg4", 8, 0, True),
                                                    ("cfg4", 16, 9, True), ("cfg4", 2, 1, True), ("cfg4_2M", 8, 7, True),
                                                    ("cfg4_2M", 4, 1, True), ("cfg4", 8, 5, False),
                                                    ("cfg4", 8, 7, False), ("cfg3", 4, 3, False), ("cfg3", 4, 0, False)])
def test_band_sharding_matches_unsharded(wl_name, world, g, banded):
    """SURVEY 8e variant (ii) on ONE GPU.  banded: the root's second FFT pass writes the spectrum as one region per
    band (psdr_set_band_layout; the halo columns by k_band_halo) and region g is handed over as it is.  Otherwise the
    root packs band g out of the device layout (psdr_pack_band: tile-major IQ lines for cfg4, the fused real layout
    for cfg3).  A second context that never runs an FFT receives it (a device copy stands in for the RCCL scatter) and
    demodulates rank g's clients (psdr_demod_batch_from_band_region / _from_band).  Bit-identical to the unsharded
    path (which test_cfg*_fullsize check against the oracle); the last band's halo wraps around the end of the
    spectrum.  The banded root also still answers psdr_read_spectrum, its own demodulation and the pyramid."""
    import torch
    from phantomsdr_amd import SpectrumEngine
    from phantomsdr_amd._lib import check
    from phantomsdr_amd.distributed import HipBandBackend, assign_clients_by_band, band_bounds, banded_bounds
    B = _bench()
    # (cfg4_2M: cfg4's stream with 2^21-point frames - 2048-bin columns, the 8-column pass-1 tiles, n = 720)
    wl = dict(B.WORKLOADS["cfg4"], fft_size=1 << 21) if wl_name == "cfg4_2M" else B.WORKLOADS[wl_name]
    N, F, nb_, is_real = wl["fft_size"], 3, 3, wl["is_real"]
    dev = torch.device("cuda", 0)
    x = synth_stream((nb_ * F + 1) * (N // 2), is_real, seed=17, fft_size=N)
    raw = quantize_raw(x, wl["fmt"], is_real)
    del x
    ring = torch.from_numpy(raw.view(np.int16)).to(dev)
    torch.cuda.synchronize()
    ncl = 64
    mk = lambda: SpectrumEngine(wl["sps"], N, is_real, input_format=wl["fmt"], max_batch=F, max_clients=ncl + 1,
                                max_waterfall_clients=1)
    engA, engR, engG = mk(), mk(), mk()
    try:
        p = engA.params
        n, R = p["audio_fft_size"], p["fft_result_size"]
        allc = B.make_clients(dict(wl, modes=("USB", "LSB", "AM", "FM")), p, seed=0x5D5D0004, count=ncl * world)
        # plus windows on the band's own edges: the first bin of the band, and one that ends in the halo
        first, bins = band_bounds(g, R, world, n)
        nxt = band_bounds(g + 1, R, world, n)[0] if g + 1 < world else R
        allc += [("USB", first, float(first), first + 40), ("AM", nxt - 20, float(nxt - 1), min(nxt + 20, R - 1)),
                 ("LSB", nxt - 41, float(nxt - 1), nxt - 1)]
        shard = assign_clients_by_band([(l, r) for _, l, _, r in allc], R, world, n)
        assert all(len(sh) > 0 for sh in shard)
        mine = [allc[i] for i in shard[g]][-ncl:]
        assert len(mine) >= 8 and mine[-3:] == allc[-3:]
        gA, gG = ([e.add_audio_client(l, m, r, mode) for mode, l, m, r in mine] for e in (engA, engG))
        hb = engA.ctx.half_frame_bytes()
        root = HipBandBackend(torch, engR.ctx, dev, ring.data_ptr(), nb_, F, 0, world, n, banded=banded)
        recv = HipBandBackend(torch, engG.ctx, dev, 0, nb_, F, g, world, n, root=-1, banded=banded)
        assert root.banded == banded and recv.banded == banded
        assert (recv.first, recv.bins) == (banded_bounds(g, R, world, n, N >> 10) if banded else (first, bins))
        assert recv.first <= first and first + bins <= recv.first + recv.bins
        if banded:  # the root serves clients of its own too, from the banded buffer
            gR = [engR.add_audio_client(l, m, r, mode) for mode, l, m, r in mine]
        for b in range(nb_):
            engA.ctx.process_batch(ring.data_ptr(), F, offset_bytes=b * F * hb)
            engA.ctx.demod_batch(b * F)
            engA.ctx.synchronize()
            with root.stream_context():
                root.forward(b)
                bands = root.pack_bands()
                if banded:
                    engR.ctx.demod_batch(b * F)
            root.synchronize()
            if banded:
                for f in range(F):
                    assert np.array_equal(engR.ctx.read_spectrum(f).view(np.uint32), engA.ctx.read_spectrum(f).view(np.uint32))
                    assert np.array_equal(engR.ctx.read_quantized(f), engA.ctx.read_quantized(f))
                for ci in range(len(mine)):
                    a1, a3 = gA[ci].read_audio(F), gR[ci].read_audio(F)
                    assert all(np.array_equal(np.asarray(u).view(np.uint32), np.asarray(v).view(np.uint32)) for u, v in zip(a1, a3))
            with recv.stream_context():
                recv.band_tensor().copy_(bands[g])
                recv.demod_band(b * F)
            recv.synchronize()
            for ci in range(len(mine)):
                a1, p1, n1 = gA[ci].read_audio(F)
                a2, p2, n2 = gG[ci].read_audio(F)
                assert np.array_equal(a1.view(np.uint32), a2.view(np.uint32)), f"batch {b} client {ci} {mine[ci]} audio"
                assert np.array_equal(p1.view(np.uint32), p2.view(np.uint32)), f"batch {b} client {ci} pwr"
                assert np.array_equal(n1, n2)
        # a window outside the band is refused and nothing runs
        other = (g + world // 2) % world
        o_first = band_bounds(other, R, world, n)[0]
        bad = engG.add_audio_client(o_first + 5, float(o_first + 5), o_first + 30, "USB")
        fn = engG.ctx.lib.psdr_demod_batch_from_band_region if banded else engG.ctx.lib.psdr_demod_batch_from_band
        rc = fn(engG.ctx.h, C.c_void_p(recv.band.data_ptr()), recv.bins, recv.first, recv.bins, F, 0)
        assert rc != 0 and b"outside the band" in engG.ctx.lib.psdr_last_error()
        del bad
    finally:
        for e in (engR, engG):
            check(e.ctx.lib.psdr_set_stream(e.ctx.h, None))
        for e in (engA, engR, engG):
            e.close()


def test_pipelined_broadcast_backend_matches_unsharded():
    """PipelinedShardedRunner + HipPipelinedBackend on ONE GPU (world = 1): transform, staged linear copy of
    the whole spectrum (psdr_pack_band over [0, R)), demodulation one step late from the staged copy
    (psdr_demod_batch_from_band) - bit-identical to psdr_process_batch + psdr_demod_batch."""
    import torch
    from phantomsdr_amd import SpectrumEngine
    from phantomsdr_amd._lib import check
    from phantomsdr_amd.distributed import HipPipelinedBackend, PipelinedShardedRunner
    B = _bench()
    wl = B.WORKLOADS["cfg4"]
    N, F, nb_ = wl["fft_size"], 3, 3
    dev = torch.device("cuda", 0)
    x = synth_stream((nb_ * F + 1) * (N // 2), False, seed=23, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    del x
    ring = torch.from_numpy(raw.view(np.int16)).to(dev)
    torch.cuda.synchronize()
    mk = lambda: SpectrumEngine(wl["sps"], N, False, input_format="s16", max_batch=F, max_clients=32, max_waterfall_clients=1)
    engA, engB = mk(), mk()
    try:
        clients = B.make_clients(dict(wl, modes=("USB", "LSB", "AM", "FM")), engA.params, seed=5, count=32)
        gA, gB = ([e.add_audio_client(l, m, r, mode) for mode, l, m, r in clients] for e in (engA, engB))
        hb = engA.ctx.half_frame_bytes()
        runner = PipelinedShardedRunner(HipPipelinedBackend(torch, engB.ctx, dev, ring.data_ptr(), nb_, F, True), None, 0, 1, F)
        want = []
        for b in range(nb_):
            engA.ctx.process_batch(ring.data_ptr(), F, offset_bytes=b * F * hb)
            engA.ctx.demod_batch(b * F)
            engA.ctx.synchronize()
            want.append([g.read_audio(F) for g in gA])
        for b in range(nb_ + 1):
            if b < nb_:
                runner.step(b)
            else:
                runner.flush()
            runner.backend.synchronize()
            if b == 0:
                continue                      # results arrive one step late
            for ci in range(len(clients)):
                a2, p2, n2 = gB[ci].read_audio(F)
                a1, p1, n1 = want[b - 1][ci]
                assert np.array_equal(a1.view(np.uint32), a2.view(np.uint32)), f"batch {b - 1} client {ci}"
                assert np.array_equal(p1.view(np.uint32), p2.view(np.uint32)) and np.array_equal(n1, n2)
    finally:
        check(engB.ctx.lib.psdr_set_stream(engB.ctx.h, None))
        for e in (engA, engB):
            e.close()
