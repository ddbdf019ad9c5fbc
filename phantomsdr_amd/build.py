"""Builds libpsdr_hip.so (gfx950) in-tree with hipcc: one object per translation unit of csrc/ (compiled in parallel),
one link.  Cross-compiles without a GPU."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# context: lifetime / Level 1 / ingest / instrumentation; pass1, pass2: the FFT pass launchers; forward: the frame loop,
# read-back, band layout, waterfall; demod: audio clients; postchain: DC blocker / AGC / int16; group: n GPUs from one
# process over RCCL; wire: packet formats
UNITS = ["context", "pass1", "pass2", "forward", "demod", "postchain", "group", "wire"]
OUT = os.path.join(_HERE, "libpsdr_hip.so")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-fPIC"]


def sources():
    inc = os.path.join(_HERE, "..", "include", "psdr.h")
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [inc]


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in sources())


def build_extension(force=False, verbose=False, out=None, defines=(), extra=()):
    """hipcc --offload-arch=gfx950 ... -> phantomsdr_amd/libpsdr_hip.so (or `out`, e.g. a tuning variant)"""
    out = out or OUT
    if not force and out == OUT and not is_stale():
        return OUT
    objdir = os.path.join(_HERE, "..", "build", "obj", os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    dflags = [f"-D{d}" for d in defines] + list(extra)

    def one(u):
        obj = os.path.join(objdir, u + ".o")
        cmd = [HIPCC] + FLAGS + dflags + ["-c", os.path.join(CSRC, u + ".hip"), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(min(len(UNITS), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(one, UNITS))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
