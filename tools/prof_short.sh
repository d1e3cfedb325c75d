#!/bin/bash
# per-kernel times of short batch calls on long impulse responses:  bash tools/prof_short.sh <tag> <streams> <blocks> [plan]
set -e -o pipefail
tag=$1; S=$2; nb=$3; plan=${4:-1}
root=$(pwd); out=$root/gpurun_out/$tag
mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- python3 "$root/tools/bench_short_calls.py" --plan $plan --streams $S --sizes $nb > "$out/log.txt" 2>&1
rm -f "$out/run_kernel_trace.csv"
python3 - "$out" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1] + "/run_kernel_stats.csv")):
    if "ohs::" in r["Name"] and int(r["Calls"]) > 50:
        print(f'{r["Name"][:52]:52s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]):9.0f} min {r["MinNs"]:>7s}')
PY
grep device_us "$out/log.txt" | cut -c1-160
