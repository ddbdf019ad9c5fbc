#!/usr/bin/env python3
"""Any rocprofv3 --pmc pass: mean counter value per launch (last three launches) of every psdr:: kernel.
usage: pmc_generic_summary.py <dir with *counter_collection.csv> <out json>"""
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
out = {k: dict({n: sum(v[-3:]) / len(v[-3:]) for n, v in c.items()}, launches=len(next(iter(c.values())))) for k, c in rows.items()}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
