#!/usr/bin/env python3
"""Instruction mix of a kernel in hipcc's device assembly (tuning aid).
   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o /tmp/psdr.s phantomsdr_amd/csrc/psdr_api.hip
   tools/isa_mix.py /tmp/psdr.s k_fft_pass1ILi1024ELi16ELi2ELi4"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
inside, c, name = False, collections.Counter(), None
for line in txt:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        if inside:
            break
        if pat in m.group(1):
            inside, name = True, m.group(1)
        continue
    if inside:
        if "s_endpgm" in line:
            c["s_endpgm"] += 1
            continue
        m = re.match(r"\s+([a-z][a-z_0-9]+)\s", line + " ")
        if m and not line.strip().startswith((".", ";")):
            c[m.group(1)] += 1
tot = sum(c.values())
grp = collections.Counter()
for k, v in c.items():
    g = ("v_pk" if k.startswith("v_pk_") else "valu" if k.startswith("v_") else "lds" if k.startswith("ds_") else
         "vmem" if k.startswith(("global_", "buffer_", "scratch_", "flat_")) else "salu/other")
    grp[g] += v
print(name, "total", tot, dict(grp))
print(sorted(c.items(), key=lambda x: -x[1])[:30])
