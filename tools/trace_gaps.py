#!/usr/bin/env python3
"""Gaps of the headline step from a rocprofv3 kernel trace CSV: idle time between consecutive k_eq_ring launches, and from the
end of a step's last EQ launch to the end of its last convolution kernel.
    rocprofv3 --kernel-trace --output-format csv -d out -o run -- python3 bench.py --no-extras --no-cpu-baseline --steps 4 --warmup 2
    python tools/trace_gaps.py out/run_kernel_trace.csv"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if "ohs::" not in name:
            continue
        short = name.split("ohs::")[1].split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
rows.sort()
eq = [(s, e) for s, e, n in rows if n == "k_eq_ring"]
# steps = groups of six EQ launches
print(f"{len(eq)} k_eq_ring launches")
for i in range(0, len(eq) - 5, 6):
    g = eq[i:i + 6]
    gaps = [(g[k + 1][0] - g[k][1]) / 1e3 for k in range(5)]
    durs = [(e - s) / 1e3 for s, e in g]
    nxt = eq[i + 6][0] if i + 6 < len(eq) else None
    convs = [(s, e, n) for s, e, n in rows if n.startswith("k_conv_p1") and s >= g[0][0] and (nxt is None or s < nxt)]
    last_conv_end = max(e for s, e, n in convs)
    tail = (last_conv_end - g[5][1]) / 1e3
    to_next = (nxt - g[5][1]) / 1e3 if nxt else float("nan")
    print(f"step {i // 6}: EQ durations {[round(d, 1) for d in durs]} us (sum {sum(durs):.1f}); gaps between EQ launches "
          f"{[round(x, 1) for x in gaps]} us (sum {sum(gaps):.1f}); last EQ end -> last conv end {tail:.1f} us; -> next step's "
          f"first EQ start {to_next:.1f} us")
