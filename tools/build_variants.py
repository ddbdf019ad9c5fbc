#!/usr/bin/env python
"""Builds ablation variants of libpsdr_hip.so under build/variants/ (they travel to the GPU box with
the snapshot; build/ is git-ignored).  usage: build_variants.py name=DEFINE[,DEFINE] ..."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "phantomsdr_amd"))
import build as B  # noqa: E402

out = os.path.join(ROOT, "build", "variants")
os.makedirs(out, exist_ok=True)


def one(spec):
    name, _, defs = spec.partition("=")
    so = os.path.join(out, f"libpsdr_{name}.so")
    B.build_extension(force=True, out=so, defines=[d for d in defs.split(",") if d])
    return so


with ThreadPoolExecutor(4) as ex:
    for so in ex.map(one, sys.argv[1:]):
        print(so)
