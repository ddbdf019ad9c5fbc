#!/usr/bin/env python3
"""Kernel timeline of the last steps of a rocprofv3 --kernel-trace run (psdr:: kernels only).
usage: trace_timeline.py <p_kernel_trace.csv> [steps=2]"""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'psdr::' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    n = re.sub(r'^void ', '', r['Kernel_Name']).split('(')[0].replace('psdr::', '')
    r['n'] = n.split('<')[0]
rows.sort(key=lambda r: r['s'])
p1 = [r for r in rows if r['n'] == 'k_fft_pass1']
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0, t1 = p1[-2 - ns]['s'], p1[-2]['s']
print(f"window of {ns} steps = {(t1 - t0) / 1e3:.0f} us")
for r in rows:
    if t0 <= r['s'] < t1:
        print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:9.1f}  q{r['Queue_Id']:>3} {r['n']}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} vgpr {r['VGPR_Count']}+{r['Accum_VGPR_Count']}")
