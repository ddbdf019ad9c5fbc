"""BASELINE.json configs[3] and configs[4] at their STATED client sets, on the one device a GPU box has (VERDICT r4
next #5; src/websocket.cpp:156-236 is the fan-out they exercise):

  configs[3]  35 MSPS IQ, 2^20-pt FFT, 256 audio clients "sharded 8 x MI355X with RCCL spectrum broadcast"
  configs[4]  70 MSPS real, 2^22-pt R2C, 1024 clients + 64 zoomed waterfalls, 8 x MI355X

* one context with the WHOLE client set (the one-GPU share of 32 / 128 clients is tests/test_gpu_fullsize.py's);
* the same client sets through psdr_group_* in the north star's sharding (clients over the ranks, spectrum exchange):
  one rank with the RCCL collectives forced, and EIGHT ranks - on the one device - with the peers pulling the spectrum
  by device copies (PSDR_SHARD_PEER_COPY): the placement (client i on rank i mod 8: 32 / 128 per rank), the exchange
  into every rank's own spectrum buffer, the per-rank demodulation and fetch, and the waterfall clients on the root,
  exactly as an 8-GPU node would run them but for the transport.
Everything against the oracle: every client's audio and power on every frame, every waterfall row."""
import numpy as np
import pytest

from helpers import quantize_raw, synth_stream
from oracle import oracle as O
from test_gpu_fullsize import _bench, _check_audio, _oracle_clients, run_workload

pytestmark = pytest.mark.gpu


def _wl(name):
    B = _bench()
    if name == "configs3":
        return dict(B.WORKLOADS["cfg4"], audio=256, waterfall=0)
    return dict(B.WORKLOADS["cfg5"], audio=1024, waterfall=64)


def test_configs4_whole_client_set_on_one_context_vs_oracle():
    """2^22-pt real, n = 720: 1024 mixed USB / LSB / AM / FM clients + 64 waterfall clients (the full span and 63 zooms at
    random levels), three frames in two batches"""
    run_workload(_wl("configs4"), splits=(2, 1))


def test_configs3_whole_client_set_on_one_context_vs_oracle():
    run_workload(_wl("configs3"), splits=(2, 1))


@pytest.mark.parametrize("ranks,how", [(1, "rccl_forced"), (8, "peer_copy")])
@pytest.mark.parametrize("name", ["configs3", "configs4"])
def test_configs_through_the_group_vs_oracle(name, ranks, how):
    from phantomsdr_amd import Group, WaterfallClient
    from phantomsdr_amd.core import derived_params
    B = _bench()
    wl = _wl(name)
    N, is_real, fmt = wl["fft_size"], wl["is_real"], wl["fmt"]
    p = derived_params(wl["sps"], N, is_real)
    R, n, levels, skip = p["fft_result_size"], p["audio_fft_size"], p["downsample_levels"], p["skip_num"]
    F, nb = 2, 2
    clients = B.make_clients(wl, p, seed=0x5D5D0004)
    waterfalls = B.make_waterfalls(wl, p, seed=0x5D5D0004)
    assert len(clients) == wl["audio"] and len(waterfalls) == wl["waterfall"]
    x = synth_stream((nb * F + 1) * (N // 2), is_real, seed=55, fft_size=N)
    raw = quantize_raw(x, fmt, is_real)
    del x
    g = Group([0] * ranks, "clients", N, is_real, levels, force_comm=how == "rccl_forced", peer_copy=how == "peer_copy",
              additional_size=n, audio_fft_size=n, input_format=fmt, max_batch=F, max_clients=(len(clients) + ranks - 1) // ranks,
              max_waterfall_clients=max(len(waterfalls), 1), skip_num=skip, waterfall_size=1024)
    try:
        root = g.root
        d = root.dev_alloc(raw.nbytes)
        root.h2d(d, raw)
        gids = [g.client_add(l, m, r, mode) for mode, l, m, r in clients]
        per_rank = np.bincount([g.client_rank(gid) for gid in gids], minlength=ranks)
        assert per_rank.tolist() == [len(clients) // ranks] * ranks, "client i lives on rank i mod n"
        gwf = []
        for lv, l, r in waterfalls:  # waterfall clients stay on the root (they read only its pyramid)
            w = WaterfallClient(root)
            w.set_waterfall_range(lv, l, r)
            gwf.append(w)
        conv = O.convert(raw, fmt)
        halves = (conv if is_real else conv.view(np.complex64)).reshape(nb * F + 1, N // 2)
        ocl = _oracle_clients(clients, is_real, n, R)
        fo = O.FFT(N, is_real, levels, 0, n)
        hb = root.half_frame_bytes()
        for b in range(nb):
            g.step(d, F, b * F, offset_bytes=b * F * hb)
            g.fetch()
            wrows = [w.read_waterfall()[0] for w in gwf]
            si = 0
            for f in range(F):
                fr = b * F + f
                fo.load(halves[fr], halves[fr + 1])
                fo.execute()
                spec_o = fo.output().copy()
                if fr % skip == 0:
                    qg = root.read_quantized(f)
                    for wi, (lv, l, r) in enumerate(waterfalls):
                        assert np.array_equal(wrows[wi][si], root.quantized_level(qg, lv)[l:r]), f"frame {fr} waterfall {wi}"
                        dq = np.abs(wrows[wi][si].astype(np.int16) - fo.quantized_level(lv)[l:r].astype(np.int16))
                        assert dq.max() <= 1 and (dq != 0).mean() <= 5e-3, f"frame {fr} waterfall {wi} vs oracle"
                    si += 1
                for ci, (gid, o) in enumerate(zip(gids, ocl)):
                    a_g, p_g, nan_g = g.fetched_audio(gid, f)
                    a_o, p_o, _, dropped = o.send_audio(spec_o, fr, fft=fo)
                    _check_audio(f"{name} x{ranks} {how} frame {fr} client {ci} {clients[ci]}", o.mode, a_g, p_g, nan_g, a_o, p_o, dropped, o)
            for wi in range(len(waterfalls)):
                assert wrows[wi].shape[0] == si
        bytes_link, ms = g.link_stats()
        assert bytes_link == F * ((N // 2 + 2) if is_real else N) * 8 and ms > 0, "the spectrum batch crossed the (stand-in) link"
        g.synchronize()
        root.dev_free(d)
    finally:
        g.close()
