"""The launch shapes bench.py TIMES, against the oracle.

test_gpu_fullsize.py runs every BASELINE shape for 3-5 frames: a fused real-input launch that small
walks chains of 1 or 2 tiles, and no frame of a 256-frame launch ever meets the oracle.  Here:

  * the fused real-input second pass (k_fft_pass2_real) with chain segments of 4, 16, 32 and 64 tiles
    at 2^21 points and of 64 tiles at 2^22 points (PSDR_SEG_LEN pins what real_seg_len() would pick
    for F = 256: 32 tiles at 2^21, 64 at 2^22) - a chain of >= 3 tiles has a MIDDLE tile, with carry-in
    AND carry-out and the LDS double-buffer toggle - spectrum, int8 pyramid, every client's audio and
    the gathered waterfall rows against the oracle                    (src/fft_impl.cpp:144-174)
  * ONE F = 256 launch each of cfg2 and cfg3, exactly as bench.py's SingleGpuRun drives it (its ring
    generator, its clients, psdr_process_batch + psdr_demod_batch + psdr_waterfall_batch), two
    consecutive batches so that the cross-batch tails are in; frames 0, 2, 127 and 255 of the SECOND
    launch - spectrum, pyramid, waterfall rows and all clients' audio - against the oracle run on the
    same raw half-frames                                             (src/signal.cpp:102-275)
"""
import os
import sys

import numpy as np
import pytest

from helpers import rel_err, rel_l2
from oracle import oracle as O
from test_gpu_fullsize import SPEC_L2, SPEC_TOL, _check_audio, _check_pyramid, _oracle_clients, run_workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _bench():
    import bench
    return bench


@pytest.fixture
def seg_len_env():
    old = os.environ.get("PSDR_SEG_LEN")

    def set_(v):
        os.environ["PSDR_SEG_LEN"] = str(v)
    yield set_
    if old is None:
        os.environ.pop("PSDR_SEG_LEN", None)
    else:
        os.environ["PSDR_SEG_LEN"] = old


@pytest.mark.parametrize("wl_name,seg_len,splits", [("cfg3", 4, (2, 1)), ("cfg3", 16, (2, 1)), ("cfg3", 32, (1, 2)),
                                                    ("cfg3", 64, (2, 1)), ("cfg5", 64, (1, 1)), ("cfg5", 128, (1, 1))])
def test_fused_real_pass_long_chains_vs_oracle(wl_name, seg_len, splits, seg_len_env):
    """chain segments as long as the bench's (and longer: one chain per frame), few frames, everything vs the oracle"""
    seg_len_env(seg_len)  # read by psdr_create
    B = _bench()
    wl = dict(B.WORKLOADS[wl_name])
    if wl_name == "cfg5":
        wl["audio"] = 32  # the full client set is test_cfg5_share_fullsize_vs_oracle's; here the chain is the subject
    run_workload(wl, splits=splits, seed=1234 + seg_len)


def test_bench_seg_len_is_what_the_chain_tests_cover():
    """what the segment plan (forward.hip: real_seg_len / seg_plan, restated in test_real_fused_model.py) picks for the
    batch sizes the bench has used: uniform 32- and 64-tile segments at 256 frames, the hand-off plan at 512"""
    import bench
    from test_real_fused_model import seg_plan
    for G, sl in ((64, 32), (128, 64)):
        tab, handoff, _ = seg_plan(G, 256)
        assert not handoff and {t[2] for t in tab} == {sl}
    # bench.DEFAULT_BATCH = 512 frames, more than one per work-group: the hand-off plan
    # (test_handoff_plan_bit_identical_to_whole_frame_segments below runs it against whole-frame segments)
    assert bench.DEFAULT_BATCH == 512 and seg_plan(64, 512)[1] and seg_plan(128, 512)[1]


@pytest.mark.parametrize("log2n", [21, 22])
def test_handoff_plan_bit_identical_to_whole_frame_segments(log2n, monkeypatch):
    """The fused real second pass with the round-4 segment plan (segments of G/4 ... 1 tiles drawn level-major by tickets,
    the carried row handed from work-group to work-group through memory behind a flag INSIDE the launch) against the same
    512-frame batches with one whole-frame segment per frame (PSDR_SEG_LEN = G: no hand-off at all): the spectrum and
    the int8 pyramid of EVERY frame of two consecutive launches (both result sets, two epochs of the flags) must be
    bit-identical - a stale carried row would show in the mirror-side octets of a segment's first tile."""
    import hashlib
    from phantomsdr_amd import Context
    N, F = 1 << log2n, 512
    G = (N // 2 // 1024) // 16
    rng = np.random.default_rng(log2n)
    raw = rng.integers(-2000, 2000, size=(2 * F + 1) * (N // 2), dtype=np.int16)
    raw[::7919] = 32767  # spikes: every tile's carried row is its own

    def run(seg_len):
        if seg_len:
            monkeypatch.setenv("PSDR_SEG_LEN", str(seg_len))
        else:
            monkeypatch.delenv("PSDR_SEG_LEN", raising=False)
        ctx = Context(N, True, 12 if log2n == 22 else 11, input_format="s16", max_batch=F)
        try:
            d = ctx.dev_alloc(raw.nbytes)
            ctx.h2d(d, raw)
            out = []
            hb = ctx.half_frame_bytes()
            ctx.process_batch(d, F)
            ctx.process_batch(d, F, offset_bytes=F * hb)  # the second launch's results are what can be read
            for f in range(F):
                out.append((hashlib.blake2b(ctx.read_spectrum(f).tobytes(), digest_size=16).digest(),
                            hashlib.blake2b(ctx.read_quantized(f).tobytes(), digest_size=16).digest()))
            ctx.process_batch(d, F)  # third launch, first half again: the other result set, a third epoch
            for f in (0, 1, F // 2, F - 1):
                out.append((hashlib.blake2b(ctx.read_spectrum(f).tobytes(), digest_size=16).digest(),
                            hashlib.blake2b(ctx.read_quantized(f).tobytes(), digest_size=16).digest()))
            if not seg_len:  # the default plan: it IS the hand-off plan, and the fallback to a seam stays the exception
                import ctypes as C
                from phantomsdr_amd import _lib
                fn = _lib.load().psdr_debug_seg_fallbacks
                fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_void_p, C.c_uint]
                ns, fb = C.c_uint(0), C.c_uint(0)
                assert fn(ctx.h, C.byref(ns), C.byref(fb), None, 0) == 0
                assert ns.value == F * (7 + log2n - 20) and fb.value < ns.value // 4, (ns.value, fb.value)
            ctx.dev_free(d)
            return out
        finally:
            ctx.close()
    a, b = run(0), run(G)
    bad_x = [f for f in range(len(a)) if a[f][0] != b[f][0]]
    bad_q = [f for f in range(len(a)) if a[f][1] != b[f][1]]
    assert not bad_x and not bad_q, (bad_x[:8], bad_q[:8])


@pytest.mark.parametrize("wl_name,post", [("cfg2", False), ("cfg3", False), ("cfg3", True), ("cfg2", True)])
def test_bench_launch_256_frames_vs_oracle(wl_name, post):
    """post: with psdr_set_post_chain(1) - the passes then leave a CU per XCD free (248 work-groups: the real plan's chain
    segments and hand-offs on another grid than the one their plan was sized for), the chain's kernels run beside them.
    bench.py's own launch: SingleGpuRun.step() twice with F = bench.DEFAULT_BATCH (512; 256 up to round 3), then
    frames {0, 2, F/2 - 1, F - 1} of the second launch against the oracle (which runs frames g-2, g-1, g: the overlap-add tail and FM's last sample are
    functions of the two preceding frames)."""
    import torch
    B = _bench()
    wl = B.WORKLOADS[wl_name]
    import bench
    F = bench.DEFAULT_BATCH
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    N, is_real = wl["fft_size"], wl["is_real"]
    hb_mib = (N // 2) * (1 if is_real else 2) * 2 / 2 ** 20
    run = B.SingleGpuRun(torch, dev, 0, wl_name, wl, F, int(2 * F * hb_mib) + 8)
    try:
        assert run.nbatches == 2
        eng, p = run.eng, run.params
        R, n, levels, skip = p["fft_result_size"], p["audio_fft_size"], p["downsample_levels"], p["skip_num"]
        if post:
            eng.ctx.set_post_chain(True)
        run.step(0)
        run.step(1)
        run.sync()
        first = F  # frame number of the second launch's first frame
        got = [g.read_audio(F) for g in eng.audio_clients]
        wrows = [w.read_waterfall()[0] for w in eng.waterfall_clients]
        sent = [f for f in range(F) if (first + f) % skip == 0]
        for wi, w in enumerate(wrows):
            assert w.shape[0] == len(sent)
        check = sorted({0, 2, F // 2 - 1, F - 1} | ({sent[0], sent[-1]} if sent and run.waterfalls else set()))
        assert any(f in sent for f in check) or not run.waterfalls
        nb = N // 2 if is_real else N
        fo = O.FFT(N, is_real, levels, 0, n)
        ring = run.ring  # [halves][samples] int16 on the device
        for f in check:
            g = first + f
            rows = ring[g - 2: g + 2].cpu().numpy()  # halves g-2 .. g+1
            conv = O.convert(rows.reshape(-1), "s16")
            halves = (conv if is_real else conv.view(np.complex64)).reshape(4, N // 2)
            ocl = _oracle_clients(run.clients, is_real, n, R)
            for k in range(3):
                fo.load(halves[k], halves[k + 1])
                fo.execute()
                spec_o = fo.output().copy()
                res = [o.send_audio(spec_o, g - 2 + k, fft=fo) for o in ocl]
            tag = f"{wl_name} frame {f} of the second {F}-frame launch"
            Xg = eng.ctx.read_spectrum(f)
            assert rel_err(Xg[:nb], spec_o[:nb]) < SPEC_TOL, tag
            assert rel_l2(Xg[:nb], spec_o[:nb]) < SPEC_L2, tag
            qg = eng.ctx.read_quantized(f)
            _check_pyramid(qg, Xg, fo.quantized().copy(), N, is_real, levels, tag)
            if f in sent:
                si = sent.index(f)
                for wi, (lv, l, r) in enumerate(run.waterfalls):
                    assert np.array_equal(wrows[wi][si], eng.ctx.quantized_level(qg, lv)[l:r]), f"{tag} waterfall {wi}"
            for ci, o in enumerate(ocl):
                a_o, p_o, _, dropped = res[ci]
                _check_audio(f"{tag} client {ci} {run.clients[ci]}", o.mode, got[ci][0][f], got[ci][1][f], got[ci][2][f],
                             a_o, p_o, dropped, o)
    finally:
        run.close()


def test_bench_gpus_2_runs_every_sharding_as_two_processes():
    """`python bench.py --gpus 2` with no launcher around it: the self-launch, two ranks, all five shardings with the
    HIP back-ends.  A one-GPU box cannot run RCCL with two ranks, so PSDR_BENCH_ONE_DEVICE=1 puts both ranks on cuda:0
    over gloo: an orchestration test (it found int16 tensors handed to a collective), not a measurement."""
    import json
    import subprocess
    env = dict(os.environ, PSDR_BENCH_ONE_DEVICE="1", PSDR_BENCH_CPU_BUDGET_S="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batch", "8", "--ring-mib", "64"], capture_output=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["testing_mode"]
    assert set(d["sharding"]) == {"clients", "clients_pipelined", "raw", "band", "time"}
    for mode, v in d["sharding"].items():
        assert v.get("error") is None and v["value"] > 0, (mode, v)
    # `value` is the north star's sharding (BASELINE.json configs[3]: clients over the GPUs, RCCL spectrum broadcast), band /
    # raw / pipelined / time sit beside it, and the C-side group leg (one child process through psdr_group_*; here: the one
    # device with forced collectives) reported its three shardings
    assert d["shard"] == "clients" and d["value"] == d["sharding"]["clients"]["value"] and d["north_star_sharding"]["value"] > 0
    assert "spectrum" in d["config"]["parallelism"] and "broadcast" in d["config"]["parallelism"]
    cg = d["c_group"]
    assert cg and "by_shard" in cg, cg
    for shard, v in cg["by_shard"].items():
        assert v.get("error") is None and v["value"] > 0, (shard, v)
        # one device: the exchange is a copy onto itself - never reported as a link rate
        assert v["GB_per_s_per_link_during_exchange"] is None and v["GB_per_s_per_link_over_the_step"] is None
    # the N > 1 line is adjudicable on its own: CPU baseline of the job and the root GPU's roofline block
    cpu = d["cpu_baseline"]
    assert cpu and cpu.get("error") is None and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["kind"] == "port" and cpu["sample"]
    rf = d["roofline"]
    assert rf and rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["target_frac"] == 0.40
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 2e-4 and rf["kernel"] and rf["kernel_frac"] > 0
    # `frac` reproduces from `value` alone: frames/s = value * 1e6 / (N/2)
    fps = d["value"] * 1e6 / (d["config"]["fft_size"] // 2)
    assert abs(rf["frac"] - rf["algorithmic_bytes_per_frame"] * fps / 8.0e12) < 2e-3 * max(rf["frac"], 1e-9) + 1e-4


def test_handoff_two_contexts_interleaved_and_640_frames(monkeypatch):
    """Two contexts on one device, each with its own streams, flags and epochs, launching 640-frame batches of 2^21-point
    real frames alternately WITHOUT synchronising in between (their second passes overlap on the chip: neither gets all
    the CUs, work-groups of both start late - the hand-off never waits, so nothing can lock up), against one context with
    whole-frame segments.  640 frames: a batch size that is no power of two and no multiple of the work-group count."""
    import hashlib
    from phantomsdr_amd import Context
    N, F = 1 << 21, 640
    rng = np.random.default_rng(77)
    raw = rng.integers(-1500, 1500, size=(F + 1) * (N // 2), dtype=np.int16)

    def digest(ctx):
        return [hashlib.blake2b(ctx.read_quantized(f).tobytes(), digest_size=12).digest() for f in range(0, F, 3)]
    monkeypatch.setenv("PSDR_SEG_LEN", "64")
    ref_ctx = Context(N, True, 11, input_format="s16", max_batch=F)
    try:
        d = ref_ctx.dev_alloc(raw.nbytes)
        ref_ctx.h2d(d, raw)
        ref_ctx.process_batch(d, F)
        ref = digest(ref_ctx)
        ref_ctx.dev_free(d)
    finally:
        ref_ctx.close()
    monkeypatch.delenv("PSDR_SEG_LEN")
    a, b = Context(N, True, 11, input_format="s16", max_batch=F), Context(N, True, 11, input_format="s16", max_batch=F)
    try:
        da, db = a.dev_alloc(raw.nbytes), b.dev_alloc(raw.nbytes)
        a.h2d(da, raw)
        b.h2d(db, raw)
        for _ in range(3):  # asynchronous launches: the two contexts' kernels share the device
            a.process_batch(da, F)
            b.process_batch(db, F)
        assert digest(a) == ref
        assert digest(b) == ref
        a.dev_free(da)
        b.dev_free(db)
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("log2n,F", [(21, 320), (21, 257), (22, 288)])
def test_handoff_plan_between_one_and_two_frames_per_work_group(log2n, F, monkeypatch):
    """Batches of more than one frame per work-group take the hand-off plan (forward.hip: seg_plan_counts).  Below two
    frames per work-group a segment's predecessor is less than two segments ahead: many first tiles fall back to a seam -
    every frame's pyramid and spectrum must still be bit-identical to whole-frame segments."""
    import ctypes as C
    import hashlib
    from phantomsdr_amd import Context, _lib
    from test_real_fused_model import seg_plan
    N = 1 << log2n
    G = (N // 2 // 1024) // 16
    assert seg_plan(G, F)[1]
    rng = np.random.default_rng(log2n * 1000 + F)
    raw = rng.integers(-2500, 2500, size=(2 * F + 1) * (N // 2), dtype=np.int16)

    def run(seg_len):
        if seg_len:
            monkeypatch.setenv("PSDR_SEG_LEN", str(seg_len))
        else:
            monkeypatch.delenv("PSDR_SEG_LEN", raising=False)
        ctx = Context(N, True, 12 if log2n == 22 else 11, input_format="s16", max_batch=F)
        try:
            d = ctx.dev_alloc(raw.nbytes)
            ctx.h2d(d, raw)
            ctx.process_batch(d, F)
            ctx.process_batch(d, F, offset_bytes=F * ctx.half_frame_bytes())
            if not seg_len:
                fn = _lib.load().psdr_debug_seg_fallbacks
                fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_void_p, C.c_uint]
                ns, fb = C.c_uint(0), C.c_uint(0)
                assert fn(ctx.h, C.byref(ns), C.byref(fb), None, 0) == 0
                assert ns.value == F * (7 + log2n - 20), "the hand-off plan ran"
            out = [(hashlib.blake2b(ctx.read_spectrum(f).tobytes(), digest_size=12).digest(),
                    hashlib.blake2b(ctx.read_quantized(f).tobytes(), digest_size=12).digest()) for f in range(F)]
            ctx.dev_free(d)
            return out
        finally:
            ctx.close()
    a, b = run(0), run(G)
    bad = [f for f in range(F) if a[f] != b[f]]
    assert not bad, bad[:8]
