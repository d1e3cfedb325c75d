#!/usr/bin/env python3
"""Summarises two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2) into
profiles/<tag>_pmc.json and profiles/pmc_latest.json (read by bench.py for `roofline.traffic`).

Units / corrections (same guide, section HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly 1/2 of the bytes of a coalesced streaming read, so fetch bytes = FETCH_SIZE * 1024 * 2;
WRITE_SIZE is exact for streaming stores.  Calibration in this workload: k_eq_pass reads its
983.56 MB input exactly once -> raw FETCH_SIZE 492.07 MB (x2 = 984.1 MB), WRITE_SIZE 983.61 MB.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag>
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "ohs::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ohs::", "").split("<")[0]
            agg[name].append(float(r["Counter_Value"]))
    return agg


def main():
    fpath, wpath, tag = sys.argv[1:4]
    f = load(fpath, "FETCH_SIZE")
    w = load(wpath, "WRITE_SIZE")
    out = {"_note": "per-launch averages; fetch corrected x2 for gfx950 (FETCH_SIZE counts 128-B requests as 64 B)",
           "_source": [os.path.basename(fpath), os.path.basename(wpath)], "_tag": tag}
    for k in sorted(set(f) | set(w)):
        fv, wv = f.get(k, [0.0]), w.get(k, [0.0])
        fb = sum(fv) / len(fv) * 1024.0 * 2.0
        wb = sum(wv) / len(wv) * 1024.0
        out[k] = {"launches_sampled": max(len(fv), len(wv)), "fetch_bytes_per_launch": round(fb),
                  "write_bytes_per_launch": round(wb), "hbm_bytes_per_launch": round(fb + wb)}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    for name in (f"{tag}_pmc.json", "pmc_latest.json"):
        with open(os.path.join(ROOT, "profiles", name), "w") as fh:
            json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
