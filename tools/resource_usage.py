#!/usr/bin/env python
"""Per-kernel register / LDS / scratch usage of the library as hipcc reports it (-Rpass-analysis=kernel-resource-usage).
usage: tools/resource_usage.py [usage.txt]   (without an argument: compiles the kernel translation units of phantomsdr_amd/csrc, ~25 s)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    txt = open(sys.argv[1]).read()
else:
    from concurrent.futures import ThreadPoolExecutor
    d = tempfile.mkdtemp()
    extra = os.environ.get("PSDR_DEFINES", "").split()  # e.g. PSDR_DEFINES="-DPSDR_HANDOFF_FULL_DRAIN"

    def one(u):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-fPIC", "-c",
                            "-Rpass-analysis=kernel-resource-usage", *extra, os.path.join(ROOT, "phantomsdr_amd", "csrc", u + ".hip"),
                            "-o", os.path.join(d, u + ".o")], capture_output=True, text=True)
        return r.stderr

    with ThreadPoolExecutor(5) as ex:
        txt = "".join(ex.map(one, ["pass1", "pass2", "forward", "demod", "postchain"]))
SCR, OCC, LDS = r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]"
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split(" [")[0]
    g = lambda k: (re.search(k + r": (\S+)", b) or [None, "?"])[1]  # noqa: E731
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(psdr::\w+.*", "", dn).replace("void psdr::", "").replace("psdr::", "")
    print("%-52s VGPR %4s AGPR %3s SGPR %4s scratch %4s occ %2s LDS %s" % (dn[:52], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(SCR), g(OCC), g(LDS)))
