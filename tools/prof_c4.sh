#!/bin/bash
# Config-4 profile on the GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) and HBM-side traffic
# (FETCH_SIZE / WRITE_SIZE in passes of their own) of tools/bench_c4_plans.py under one plan.
#   gpurun -- 'bash tools/prof_c4.sh <tag> <plan> [extra bench args]'   -> gpurun_out/<tag>/
set -e -o pipefail
tag=${1:-prof_c4}; plan=${2:-2}; shift 2 || true
root=$(pwd); out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
P=(python3 "$root/tools/bench_c4_plans.py" --plans "$plan" --steps 20 "$@")
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o run -- "${P[@]}" > "$out/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o run -- "${P[@]}" > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o run -- "${P[@]}" > "$out/write.log" 2>&1
find "$out" -name "*_kernel_trace.csv" -delete
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats (ns)")
    for r in csv.DictReader(open(f)):
        print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]):10.0f} min {r["MinNs"]:>8s} max {r["MaxNs"]:>8s} pct {r["Percentage"]}')
for what in ("fetch", "write"):
    for f in glob.glob(out + f"/{what}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            a = agg[r["Kernel_Name"][:64]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
        print(f"== {what} (counter units per launch; FETCH_SIZE / WRITE_SIZE are KiB-ish units: see MI355X_MICROARCH.md)")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{k:64s} launches {n:5d} per launch {v / n:14.1f}")
PY
