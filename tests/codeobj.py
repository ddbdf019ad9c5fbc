"""Reads the gfx950 code objects out of the built libpsdr_hip.so (the clang offload bundles of its .hip_fatbin section) and
their kernel metadata (llvm-readelf --notes): what the loader will actually allocate per wave.  Test infrastructure."""
import os
import re
import struct
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "phantomsdr_amd", "libpsdr_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so=SO, arch="gfx950"):
    b = open(so, "rb").read()
    out, pos = [], 0
    while True:
        i = b.find(MAGIC, pos)
        if i < 0:
            return out
        (nb,) = struct.unpack_from("<Q", b, i + 24)
        o = i + 32
        for _ in range(nb):
            off, size, tl = struct.unpack_from("<QQQ", b, o)
            triple = b[o + 24:o + 24 + tl].decode()
            o += 24 + tl
            if arch in triple and size:
                out.append(b[i + off:i + off + size])
        pos = i + len(MAGIC)


def kernel_metadata(so=SO):
    """{demangled kernel name: {"vgpr": .., "agpr": .., "sgpr": .., "scratch": .., "lds": .., "wg": ..}}"""
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for k, co in enumerate(code_objects(so)):
            p = os.path.join(d, f"co{k}.elf")
            open(p, "wb").write(co)
            txt = subprocess.run([READELF, "--notes", p], capture_output=True, text=True, check=True).stdout
            for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
                blk = ".agpr_count:" + blk
                g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))  # noqa: E731
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                res[name] = {"agpr": g("agpr_count"), "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"),
                             "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size"),
                             "wg": g("max_flat_workgroup_size")}
    names = list(res)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return {re.sub(r"^void ", "", dn): res[n] for n, dn in zip(names, dem)}
