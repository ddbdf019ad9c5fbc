"""Builds libpsdr_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "psdr_api.hip")
OUT = os.path.join(_HERE, "libpsdr_hip.so")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-shared", "-fPIC"]


def sources():
    d = os.path.join(_HERE, "csrc")
    inc = os.path.join(_HERE, "..", "include", "psdr.h")
    return [os.path.join(d, f) for f in sorted(os.listdir(d))] + [inc]


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in sources())


def build_extension(force=False, verbose=False, out=None, defines=()):
    """hipcc --offload-arch=gfx950 ... -> phantomsdr_amd/libpsdr_hip.so"""
    out = out or OUT
    if not force and out == OUT and not is_stale():
        return OUT
    cmd = [HIPCC] + FLAGS + [f"-D{d}" for d in defines] + ["-o", out, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
