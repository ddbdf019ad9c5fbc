"""What the reference does NOT do to a client's state, reproduced on the GPU and checked against the oracle:

* a client whose socket is backed up gets no send_audio call (src/websocket.cpp:170-176): overlap-add tails, FM's last
  sample, DC blocker and AGC stand still for the frame (src/signal.cpp:200-203, 273-284) - psdr_client_set_paused;
* a USB / LSB frame dropped by the NaN guard throws BEFORE audio_real_prev is replaced (src/signal.cpp:266-275): the
  next frame adds the tail of the last frame that survived.  AM / FM state moves before the guard (:200-203)."""
import ctypes as C

import numpy as np
import pytest

from helpers import check_fm, pwr_tolerance, quantize_raw, rel_l2, synth_stream
from oracle import oracle as O
from test_gpu_parity import levels_for

pytestmark = pytest.mark.gpu


def _compare(a_g, a_o, o, tag):
    if o.mode == O.FM:
        check_fm(a_g, a_o, o.baseband()[: o.n // 2], o.bb_prev, tag, fwd_scale=max(o.fwd_scale, o.fwd_scale_prev))
    else:
        assert rel_l2(a_g, a_o) < 1e-4, f"{tag}: rel L2 {rel_l2(a_g, a_o):.2e}"


@pytest.mark.parametrize("n,F,is_real", [(360, 6, 0), (248, 5, 0), (720, 9, 1), (360, 1, 0)])
def test_paused_clients_keep_their_state_like_the_reference(n, F, is_real):
    """Batches of F frames; clients sit out whole batches (psdr_client_set_paused).  The oracle's clients get no
    send_audio call for those frames.  Float audio against the oracle, PCM (post chain on) bit-exact against the
    oracle's chain fed with the GPU's float audio of the frames that were NOT skipped; a paused client's results read
    as PSDR_ERR_NO_DATA.  n = 360 / 720: the chain kernel, 248: the generic kernels."""
    from phantomsdr_amd import AudioClient, Context, PsdrError
    N = 1 << 15
    R = N // 2 if is_real else N
    nb = 10 if F > 1 else 40
    levels = levels_for(R)
    x = synth_stream((nb * F + 1) * (N // 2), bool(is_real), seed=91, fft_size=N)
    raw = quantize_raw(x, "s16", bool(is_real))
    conv = O.convert(raw, "s16")
    halves = (conv if is_real else conv.view(np.complex64)).reshape(nb * F + 1, N // 2)
    am = int(0.11 * N) if is_real else int((0.11 * N - (N // 2 + 1)) % N)
    fm = int(0.31 * N) if is_real else int((-0.21 * N - (N // 2 + 1)) % N)
    specs = [("USB", am, float(am), am + 80), ("LSB", am - 80, am + 0.5, am), ("AM", am - 100, float(am), am + 100),
             ("FM", fm - 100, fm + 0.5, fm + 100), ("AM", fm - 90, float(fm + 1), fm + 90)]
    # batch -> paused clients (client 4 is paused before its first batch ever runs; client 3 twice in a row and again)
    sched = {0: {4}, 1: {4, 1}, 2: {1, 3}, 3: {3}, 5: {0, 2}, 6: {3}, 8: {0, 1, 2, 3, 4}}
    if F == 1:
        sched = {b: set(c for c in range(5) if (b * 7 + c * 3) % 11 < 3) for b in range(nb)}
    ctx = Context(N, bool(is_real), levels, additional_size=n, audio_fft_size=n, audio_rate=12000, input_format="s16",
                  max_batch=F, max_clients=len(specs))
    try:
        ctx.set_post_chain(True)
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        gcl, ocl, chains = [], [], []
        for mode, l, mid, r in specs:
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            g.set_audio_range(l, mid, r)
            gcl.append(g)
            o = O.AudioClient(bool(is_real), n, 12000, R)
            o.set_audio_demodulation(mode)
            o.set_audio_range(l, mid, r)
            ocl.append(o)
            chains.append(O.PostChain(12000))
        fo = O.FFT(N, bool(is_real), levels, 0, n)
        hb = ctx.half_frame_bytes()
        live = 0
        for b in range(nb):
            paused = sched.get(b, set())
            if b == 4:  # a mode change while nobody is paused, and one (client 1, batch 7) right after a pause
                gcl[2].set_audio_demodulation("FM"), ocl[2].set_audio_demodulation("FM"), chains[2].reset_agc()
            if b == 7:
                gcl[1].set_audio_demodulation("AM"), ocl[1].set_audio_demodulation("AM"), chains[1].reset_agc()
            for ci, g in enumerate(gcl):
                g.set_paused(ci in paused)
            ctx.process_batch(d, F, offset_bytes=b * F * hb)
            ctx.demod_batch(b * F)
            got = {}
            for ci, g in enumerate(gcl):
                if ci in paused:
                    with pytest.raises(PsdrError) as e:
                        g.read_audio(F)
                    assert e.value.code == -7
                    with pytest.raises(PsdrError) as e:
                        g.read_pcm(F)
                    assert e.value.code == -7
                else:
                    got[ci] = g.read_audio(F) + (g.read_pcm(F),)
            for f in range(F):
                frame = b * F + f
                fo.load(halves[frame], halves[frame + 1])
                fo.execute()
                spec = fo.output().copy()
                for ci, o in enumerate(ocl):
                    if ci in paused:
                        continue  # src/websocket.cpp:174-176: no send_audio call
                    a_o, p_o, _, dropped = o.send_audio(spec, frame, fft=fo)
                    a_g, p_g, nan_g, pcm_g = (got[ci][k][f] for k in range(4))
                    tag = f"client {ci} {specs[ci][0]} batch {b} frame {f} (paused this batch: {sorted(paused)})"
                    assert not dropped and nan_g == 0, tag
                    assert abs(p_g - p_o) <= pwr_tolerance(p_o, o.fwd_scale), tag
                    _compare(a_g, a_o, o, tag)
                    want = chains[ci].process(a_g)
                    assert np.array_equal(pcm_g, want), f"{tag}: PCM {np.count_nonzero(pcm_g != want)} of {want.size} differ"
                    live += int(np.count_nonzero(want) > 0)
        assert live > 20, "the AGC never opened"
        ctx.dev_free(d)
    finally:
        ctx.close()


@pytest.mark.parametrize("n,F,chain", [(360, 12, 1), (360, 12, 0), (720, 7, 1), (248, 6, 1)])
def test_ssb_tail_survives_dropped_frames_like_the_reference(n, F, chain, monkeypatch):
    """NaN samples in the INPUT (f32 format) poison the two frames that contain the half-frame - for every client.
    USB / LSB: the reference keeps audio_real_prev from the last frame that survived (the guard throws before
    src/signal.cpp:273-275), so the frame after the dropped ones adds THAT tail; AM / FM: the baseband buffers were
    already replaced (:200-203), the frame after a dropped one is poisoned through its overlap half and dropped too.
    Poisoned halves sit inside a chain, across a chain boundary (K = 4: frames 3|4, 7|8), at the batch start and at the
    batch end (the carried state).  Oracle = the reference's flow, frame by frame."""
    from phantomsdr_amd import AudioClient, Context
    monkeypatch.setenv("PSDR_DEMOD_CHAIN", str(chain))
    monkeypatch.setenv("PSDR_DEMOD_K", "4")
    N, nb = 1 << 14, 4
    levels = levels_for(N)
    nframes = nb * F
    x = synth_stream((nframes + 1) * (N // 2), False, seed=5, fft_size=N).astype(np.complex64)
    halves = x.reshape(nframes + 1, N // 2).copy()
    # half h is part of frames h-1 and h
    bad_halves = [2, 4, 5, 8, F, 2 * F - 1 + 1, 3 * F + 3, 3 * F + 4, 3 * F + 5]
    for h in bad_halves:
        halves[h, 17] = np.nan
    bad_frames = sorted({h - 1 for h in bad_halves if h >= 1} | {h for h in bad_halves if h < nframes})
    raw = halves.reshape(-1).view(np.float32).copy()
    am = int((0.11 * N - (N // 2 + 1)) % N)
    specs = [("USB", am, float(am), am + 80), ("LSB", am - 80, am + 0.5, am), ("AM", am - 100, float(am), am + 100),
             ("FM", am - 100, am + 0.5, am + 100)]
    ctx = Context(N, False, levels, additional_size=n, audio_fft_size=n, audio_rate=12000, input_format="f32",
                  max_batch=F, max_clients=len(specs))
    try:
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        gcl, ocl = [], []
        for mode, l, mid, r in specs:
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            g.set_audio_range(l, mid, r)
            gcl.append(g)
            o = O.AudioClient(False, n, 12000, N)
            o.set_audio_demodulation(mode)
            o.set_audio_range(l, mid, r)
            ocl.append(o)
        fo = O.FFT(N, False, levels, 0, n)
        hb = ctx.half_frame_bytes()
        ndropped = [0] * len(specs)
        nrecovered = 0
        for b in range(nb):
            ctx.process_batch(d, F, offset_bytes=b * F * hb)
            ctx.demod_batch(b * F)
            got = [g.read_audio(F) for g in gcl]
            for f in range(F):
                frame = b * F + f
                fo.load(halves[frame], halves[frame + 1])
                fo.execute()
                spec = fo.output().copy()
                for ci, o in enumerate(ocl):
                    a_o, p_o, _, dropped = o.send_audio(spec, frame, fft=fo)
                    a_g, nan_g = got[ci][0][f], got[ci][2][f]
                    tag = f"client {ci} {specs[ci][0]} frame {frame} (poisoned frames {bad_frames})"
                    assert bool(nan_g) == bool(dropped), tag
                    if dropped:
                        ndropped[ci] += 1
                        continue
                    _compare(a_g, a_o, o, tag)
                    nrecovered += int(ci < 2 and frame - 1 in bad_frames)
        # every poisoned frame is dropped for everybody; the complex modes lose the frame after a poisoned run as well
        assert ndropped[0] == ndropped[1] == len(bad_frames)
        assert ndropped[2] > ndropped[0] and ndropped[3] > ndropped[0]
        assert nrecovered >= 4  # SSB frames right after a dropped one matched the oracle: the surviving tail was added
        ctx.dev_free(d)
    finally:
        ctx.close()
