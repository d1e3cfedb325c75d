#!/usr/bin/env python3
"""Per-kernel averages of every counter found in rocprofv3 `*_counter_collection.csv` files.

usage: python tools/pmc_table.py <name> <csv> [<csv> ...]     -> profiles/<name>_counters.json
Per-dispatch values are averaged over the dispatches of each ohs:: kernel.  FETCH_SIZE / WRITE_SIZE (KiB) are
turned into bytes with the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (fetch x2).  SQ_* cycle
counters (WAVE_CYCLES, WAIT_*, ACTIVE_INST_*) count quad-cycles summed over waves (same guide)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kname(full):
    return full.split("(")[0].replace("void ", "").replace("ohs::", "").split("<")[0].strip()


def main():
    name, paths = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            if "ohs::" not in r["Kernel_Name"]:
                continue
            agg[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"_note": "per-dispatch averages; SQ cycle counters are quad-cycles summed over waves; "
                    "hbm bytes = FETCH_SIZE KiB x1024 x2 (gfx950) + WRITE_SIZE KiB x1024",
           "_sources": [os.path.relpath(p, ROOT) if p.startswith(ROOT) else p for p in paths]}
    for k in sorted(agg):
        row = {"dispatches": max(len(v) for v in agg[k].values())}
        for c in sorted(agg[k]):
            v = agg[k][c]
            row[c] = round(sum(v) / len(v), 1)
        if "FETCH_SIZE" in row:
            row["fetch_bytes"] = round(row["FETCH_SIZE"] * 2048.0)
        if "WRITE_SIZE" in row:
            row["write_bytes"] = round(row["WRITE_SIZE"] * 1024.0)
        if "fetch_bytes" in row and "write_bytes" in row:
            row["hbm_bytes"] = row["fetch_bytes"] + row["write_bytes"]
        wc = row.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if c in row:
                    row[c + "_frac_of_wave_cycles"] = round(row[c] / wc, 4)
        if row.get("SQ_WAVES") and row.get("SQ_INSTS_VALU"):
            row["valu_per_wave"] = round(row["SQ_INSTS_VALU"] / row["SQ_WAVES"], 1)
            if "SQ_INSTS_LDS" in row:
                row["lds_per_wave"] = round(row["SQ_INSTS_LDS"] / row["SQ_WAVES"], 1)
        if row.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_conflict_frac"] = round(row.get("SQ_LDS_BANK_CONFLICT", 0.0) / row["SQ_LDS_IDX_ACTIVE"], 4)
        out[k] = row
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    path = os.path.join(ROOT, "profiles", f"{name}_counters.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
    print("wrote", path)


if __name__ == "__main__":
    main()
