#!/usr/bin/env python3
"""L2 (TCC) hit / miss counts per kernel launch from a rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum pass.
usage: pmc_tcc_summary.py <dir with p_counter_collection.csv> <out json>"""
import collections
import csv
import glob
import json
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "psdr::" in r["Kernel_Name"]:
            rows[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in rows.items():
    hit = c.get("TCC_HIT_sum", [0])
    miss = c.get("TCC_MISS_sum", [0])
    h, m = sum(hit[-3:]) / max(len(hit[-3:]), 1), sum(miss[-3:]) / max(len(miss[-3:]), 1)
    out[k] = {"TCC_HIT_sum": h, "TCC_MISS_sum": m, "hit_rate": round(h / (h + m), 4) if h + m else None,
              "launches": len(hit)}
json.dump({"note": "per launch (mean of the last three), rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum; requests are "
                   "128-byte-line requests of the L2 (MI355X_MICROARCH.md, L2)", "kernels": out}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
