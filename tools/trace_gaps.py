#!/usr/bin/env python
"""Timeline of the FFT passes from a rocprofv3 --kernel-trace CSV: kernel durations and the idle
gaps between consecutive pass kernels (pass 1 -> pass 2 -> next pass 1).
  python tools/trace_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
p = [(s, e, "p1" if "pass1" in n else "p2" if "pass2" in n else "2p") for s, e, n in rows
     if "k_fft_pass" in n or "two_phase" in n]
p = p[len(p) // 3:]  # skip warm-up
gaps = {}
durs = {}
for a, b in zip(p, p[1:]):
    gaps.setdefault(a[2] + "->" + b[2], []).append((b[0] - a[1]) / 1e3)
for s, e, k in p:
    durs.setdefault(k, []).append((e - s) / 1e3)
for k, v in durs.items():
    print("kernel", k, "n", len(v), "avg us", round(sum(v) / len(v), 1))
for k, v in gaps.items():
    v.sort()
    print("gap", k, "n", len(v), "avg us", round(sum(v) / len(v), 1), "median", round(v[len(v) // 2], 1), "max", round(v[-1], 1))
others = {}
for s, e, n in rows:
    if "k_fft" not in n:
        others.setdefault(n.split("(")[0][:40], []).append((e - s) / 1e3)
for k, v in others.items():
    print("side", k, "n", len(v), "avg us", round(sum(v) / len(v), 1))
