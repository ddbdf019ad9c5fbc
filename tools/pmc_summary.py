#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into per-kernel HBM traffic per launch.

usage: pmc_summary.py <dir with p_counter_collection.csv for FETCH_SIZE> <dir for WRITE_SIZE>
                      <workload name> <out pmc json> <out traffic json> [frames per launch]

FETCH_SIZE / WRITE_SIZE are KiB.  Corrections (MI355X_MICROARCH.md, "HBM"): on gfx950
FETCH_SIZE tallies the 128-B requests of wide coalesced reads (16 B/lane) at 64 B, so it is
doubled for the kernels whose reads are of that kind (pass 1's raw image reads, pass 2's
float4 reads of Y); WRITE_SIZE is taken as reported.  Only steady-state launches (the most
common grid of each kernel) are averaged.
"""
import collections
import csv
import json
import sys

WIDE_READ = ("k_fft_pass1", "k_fft_pass2", "k_fft_fused", "k_untangle_real")
SHORT = {"k_fft_fused": "fft_fused", "k_fft_pass1": "fft_pass1", "k_fft_pass2": "fft_pass2", "k_untangle_real": "untangle_real",
         "k_pyramid_tail": "pyramid_tail", "k_col_tail": "col_tail", "k_real_seam": "real_seam", "k_demod_idft": "demod_idft", "k_demod_chain": "demod_chain", "k_demod_ola": "demod_ola",
         "k_waterfall_gather": "waterfall_gather", "k_untile_q": "untile_q"}


def read(dirname, counter):
    rows = collections.defaultdict(list)
    with open(dirname + "/p_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            rows[r["Kernel_Name"]].append((r["Grid_Size"], float(r["Counter_Value"])))
    out = {}
    for k, v in rows.items():
        grid = collections.Counter(g for g, _ in v).most_common(1)[0][0]
        vals = [x for g, x in v if g == grid]
        out[k] = sum(vals) / len(vals)
    return out


def main():
    fdir, wdir, wl, out_pmc, out_traffic = sys.argv[1:6]
    frames = int(sys.argv[6]) if len(sys.argv) > 6 else 64
    fetch, write = read(fdir, "FETCH_SIZE"), read(wdir, "WRITE_SIZE")
    kernels, traffic = {}, {}
    for k in sorted(set(fetch) | set(write)):
        if "psdr::" not in k:
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        wide = any(s in k for s in WIDE_READ)
        corrected = (2 * f if wide else f) * 1024 + w * 1024
        kernels[k.split("(")[0]] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                                    "fetch_doubled": wide, "hbm_bytes_per_launch": int(corrected)}
        for s, short in SHORT.items():
            if s in k:
                traffic[short] = int(corrected)
    json.dump({"note": __doc__.split("usage")[0].strip() + " Corrections: see tools/pmc_summary.py.",
               "workload": wl, "frames_per_launch": frames, "kernels": kernels}, open(out_pmc, "w"), indent=1)
    try:
        allt = json.load(open(out_traffic))
    except Exception:
        allt = {}
    traffic["_frames_per_launch"] = frames  # bench.py scales to its own batch size
    allt[wl] = traffic
    json.dump(allt, open(out_traffic, "w"), indent=1)
    print(json.dumps(traffic))


if __name__ == "__main__":
    main()
