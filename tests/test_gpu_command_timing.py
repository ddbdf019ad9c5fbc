"""When does a command take effect?  (src/websocket.cpp:156-185, src/client.cpp:19-118)

In the reference a window / mode message changes the AudioClient's fields on a websocket thread; the frame loop's NEXT
send_audio call - one frame - uses them.  Here a command changes the slot (psdr_client_set_audio_range / _demodulation)
and the next psdr_demod_batch uses it for ALL frames of its batch:

* F = 1 (what HipFanout / hip_level2.h run live): a change between two batches lands on EXACTLY the reference's frame -
  every frame before it is demodulated with the old window / mode, every frame from it on with the new one, and the
  overlap-add state is carried across the change as the reference carries it (src/signal.cpp:81-94, 316-328 reset
  nothing but the AGC);
* F > 1 (throughput runs: the bench's 512): the same change lands on the next BATCH boundary - up to F - 1 frames later
  than in the reference.  Documented in INTEGRATION.md ("Command timing"); this test pins the rule itself."""
import numpy as np
import pytest

from helpers import check_fm, pwr_tolerance, quantize_raw, rel_l2, synth_stream
from oracle import oracle as O
from test_gpu_parity import levels_for

pytestmark = pytest.mark.gpu


def _cmp(a_g, a_o, o, tag):
    if o.mode == O.FM:
        check_fm(a_g, a_o, o.baseband()[: o.n // 2], o.bb_prev, tag, fwd_scale=max(o.fwd_scale, o.fwd_scale_prev))
    else:
        assert rel_l2(a_g, a_o) < 1e-4, f"{tag}: rel L2 {rel_l2(a_g, a_o):.2e}"


@pytest.mark.parametrize("F", [1, 4])
@pytest.mark.parametrize("n,is_real", [(360, 0), (248, 1)])
def test_a_command_between_two_batches_lands_on_the_next_batchs_first_frame(n, is_real, F):
    from phantomsdr_amd import AudioClient, Context
    N = 1 << 15
    R = N // 2 if is_real else N
    nframes = 16  # (F = 4: the commands before frames 5, 7 and 9 land on the batches that start at frames 8 and 12)
    levels = levels_for(R)
    x = synth_stream((nframes + 1) * (N // 2), bool(is_real), seed=5, fft_size=N)
    raw = quantize_raw(x, "s16", bool(is_real))
    conv = O.convert(raw, "s16")
    halves = (conv if is_real else conv.view(np.complex64)).reshape(nframes + 1, N // 2)
    am = int(0.11 * N) if is_real else int((0.11 * N - (N // 2 + 1)) % N)
    fm = int(0.31 * N) if is_real else int((-0.21 * N - (N // 2 + 1)) % N)
    # (client, the frame the command arrives BEFORE, command)
    before = [("USB", am, float(am), am + 70), ("AM", am - 90, float(am), am + 90), ("LSB", fm - 70, fm + 0.5, fm)]
    commands = [(0, 5, ("window", (am + 3, am + 3.5, am + 73))), (1, 5, ("mode", "FM")), (1, 7, ("window", (fm - 90, float(fm), fm + 90))),
                (2, 9, ("mode", "USB")), (2, 9, ("window", (fm, float(fm), fm + 70)))]
    ctx = Context(N, bool(is_real), levels, additional_size=n, audio_fft_size=n, input_format="s16", max_batch=F, max_clients=len(before))
    try:
        d = ctx.dev_alloc(raw.nbytes)
        ctx.h2d(d, raw)
        gcl, ocl = [], []
        for mode, l, m, r in before:
            g = AudioClient(ctx)
            g.set_audio_demodulation(mode)
            g.set_audio_range(l, m, r)
            gcl.append(g)
            o = O.AudioClient(bool(is_real), n, 12000, R)
            o.set_audio_demodulation(mode)
            o.set_audio_range(l, m, r)
            ocl.append(o)
        fo = O.FFT(N, bool(is_real), levels, 0, n)
        hb = ctx.half_frame_bytes()

        def apply(objs, upto, done):
            """the commands that have arrived before frame `upto`"""
            for k, (ci, at, (what, arg)) in enumerate(commands):
                if k in done or at > upto:
                    continue
                done.add(k)
                for obj in objs:
                    if what == "mode":
                        obj[ci].set_audio_demodulation(arg)
                    else:
                        obj[ci].set_audio_range(*arg)
        done_g, done_o = set(), set()
        for b in range(nframes // F):
            first = b * F
            # the server's threads have delivered every command that arrived before this batch's LAST frame by the time the
            # next batch is enqueued; the GPU applies a command at the first batch boundary at or after its arrival
            apply([gcl], first, done_g)
            ctx.process_batch(d, F, offset_bytes=first * hb)
            ctx.demod_batch(first)
            got = [g.read_audio(F) for g in gcl]
            for f in range(F):
                frame = first + f
                # the reference: a command that arrived before frame `frame` is used by that frame's send_audio.
                # F = 1: the same frame.  F > 1: the GPU's rule - the batch's first frame decides for the whole batch.
                apply([ocl], frame if F == 1 else first, done_o)
                fo.load(halves[frame], halves[frame + 1])
                fo.execute()
                spec = fo.output().copy()
                for ci, o in enumerate(ocl):
                    a_o, p_o, _, dropped = o.send_audio(spec, frame, fft=fo)
                    assert not dropped and got[ci][2][f] == 0
                    tag = f"F={F} client {ci} frame {frame}"
                    assert abs(got[ci][1][f] - p_o) <= pwr_tolerance(p_o, o.fwd_scale), tag
                    _cmp(got[ci][0][f], a_o, o, tag)
        assert done_g == done_o == set(range(len(commands)))
        if F == 1:  # ... and every command landed on its own frame: the frame before it still used the old setting
            assert [at for _, at, _ in commands] == [5, 5, 7, 9, 9]
        ctx.dev_free(d)
    finally:
        ctx.close()
