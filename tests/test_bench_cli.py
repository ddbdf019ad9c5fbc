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
