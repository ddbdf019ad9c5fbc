"""bench.py's launch contract: `python bench.py --gpus N` is an N-rank run or no run at all."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _visible():
    import bench
    return bench.visible_hip_devices()


def _json_lines(out):
    return [ln for ln in out.decode().splitlines() if ln.startswith("{")]


def test_gpus_n_with_fewer_devices_exits_nonzero_and_prints_no_line():
    """the driver's plain command line: with fewer than N devices there must be no JSON line at all - in
    particular not a single-GPU line that says n_gpus: 1"""
    if _visible() >= 2:
        pytest.skip("two or more HIP devices visible: the command would run the real 2-rank bench")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, cwd=ROOT, timeout=300, env={k: v for k, v in os.environ.items()
                                                                        if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0
    assert not _json_lines(r.stdout)
    assert b"--gpus 2" in r.stderr and b"HIP device" in r.stderr


def test_gpus_n_under_a_launcher_with_another_world_size_refuses():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, cwd=ROOT, timeout=300, env=env)
    assert r.returncode != 0 and not _json_lines(r.stdout)
    assert b"refusing" in r.stderr


def test_self_launch_command_line():
    """what `--gpus N` re-executes: one process per GPU under torch.distributed.run on 127.0.0.1"""
    import bench
    calls = {}

    def fake_call(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return 0
    real_call, real_vis, real_argv = subprocess.call, bench.visible_hip_devices, sys.argv
    try:
        subprocess.call = fake_call
        bench.visible_hip_devices = lambda: 8
        sys.argv = ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"]
        with pytest.raises(SystemExit) as e:
            bench.self_launch(4)
        assert e.value.code == 0
    finally:
        subprocess.call, bench.visible_hip_devices, sys.argv = real_call, real_vis, real_argv
    cmd = calls["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_roofline_frac_is_the_surveys_path_figure_and_reproduces_from_value():
    """`roofline.frac` = B_frame x frames/s / 8e12 - derivable from
    `value` and the workload alone; the dominant kernel's own-bytes figure is `kernel_frac`, beside it."""
    import bench
    from phantomsdr_amd.core import derived_params
    wl = bench.WORKLOADS["cfg2"]
    p = derived_params(wl["sps"], wl["fft_size"], wl["is_real"])
    cl = bench.make_clients(wl, p, seed=0x5D5D0002)
    wf = bench.make_waterfalls(wl, p, seed=0x5D5D0002)
    ab = bench.algorithmic_bytes_per_frame(wl, p, cl, wf)
    value = 102_300.0  # MSamples/s (round 4's driver line)
    fps = value * 1e6 / (wl["fft_size"] // 2)
    rf = bench.path_roofline(ab["total"], fps)
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and rf["target_frac"] == 0.40
    assert abs(rf["frac"] - ab["total"] * fps / 8.0e12) < 1e-4 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert 0.355 < rf["frac"] < 0.362  # 14.70 MB x 195 k frames/s = 2.87 TB/s (VERDICT r4's recomputation: 0.359)


def test_recorded_driver_line_reproduces_its_path_fraction():
    """a recorded round-4 line (profiles/r04c_cfg2_bench.json): path.frac_of_hbm_peak (now the headline `frac`) from `value` alone"""
    import json
    import bench
    d = json.loads(open(os.path.join(ROOT, "profiles", "r04c_cfg2_bench.json")).read().strip().splitlines()[-1])
    fps = d["value"] * 1e6 / (d["config"]["fft_size"] // 2)
    rf = bench.path_roofline(d["path"]["algorithmic_bytes_per_frame"], fps)
    assert abs(rf["frac"] - d["path"]["frac_of_hbm_peak"]) < 2e-4


def test_n_gt_1_defaults():
    """the N > 1 line: `value` = the north star's sharding unless --shard says otherwise; the line carries cpu_baseline"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'main_mode = args.shard or "clients"' in src
    assert '"cpu_baseline": cpu,' in src and '"cpu_baseline": None' not in src


def test_recorded_round6_line_carries_the_served_end_and_its_arithmetic_holds():
    """the default line has `with_fetch` (16 clients) and `clients256.with_fetch` - the same step followed by
    psdr_fetch_begin / _end - and what they state is consistent: bytes per step = clients x F x (n/2 x 4 + 8) + waterfall rows,
    sustained GB/s = bytes / step, the cost ratio = the two steps' ratio (profiles/r06_bench_default.json)."""
    import json
    d = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_default.json")).read().strip().splitlines()[-1])
    F, h = d["config"]["frames_per_step"], d["config"]["audio_fft_size"] // 2
    for blk, ncl in ((d["with_fetch"], 16), (d["clients256"]["with_fetch"], 256)):
        assert blk.get("error") is None and blk["audio_clients"] == ncl and blk["frames_per_step"] == F
        for key in ("float_audio", "post_chain_pcm"):
            b = blk[key]
            assert b["d2h_bytes_per_step"] >= ncl * F * (h * 4 + 8)                       # + the waterfall rows
            assert b["d2h_bytes_per_step"] < ncl * F * (h * 4 + 8) + 4 * F * 1024 + 1     # (4 waterfall clients, at most 1024 bytes a row)
            assert abs(b["d2h_GB_per_s_sustained"] - b["d2h_bytes_per_step"] / (b["ms_per_step"] * 1e-3) / 1e9) < 0.02
            assert abs(b["over_step_without_fetch"] - b["ms_per_step"] / b["step_without_fetch_ms"]) < 2e-3
            assert b["ms_per_step"] >= 0.995 * b["step_without_fetch_ms"]                # fetching is never cheaper than not fetching
        # the float audio of a batch is ready right behind its demodulation: within a few per cent of the plain step
        assert blk["float_audio"]["over_step_without_fetch"] < 1.08
