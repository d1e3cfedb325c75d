#!/bin/bash
# Round-end evidence on the GPU box:  gpurun --timeout 1100 -- 'bash tools/collect_evidence.sh r01_v5'
# Writes under gpurun_out/<tag>/ ; afterwards, locally:
#   python tools/pmc_summary.py gpurun_out/<tag>/pmc_fetch/*/*_counter_collection.csv \
#                               gpurun_out/<tag>/pmc_write/*/*_counter_collection.csv <tag>
#   cp gpurun_out/<tag>/bench.json profiles/<tag>_bench.json ; cp .../*_kernel_stats.csv profiles/<tag>_kernel_stats.csv
# PMC passes are separate runs, without any trace domain (pool rule), the program itself after `--`.
set -e -o pipefail
tag=${1:-evidence}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --steps 10 --warmup 3 > "$out/bench.json" 2> "$out/bench.err"
echo "bench done"; cat "$out/bench.json"
python3 tools/bench_configs.py > "$out/other_configs.jsonl" 2> "$out/other_configs.err"
echo "other configs done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o run -- python3 "$root/bench.py" --no-cpu-baseline --steps 5 --warmup 2 > "$out/prof.log" 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o run -- python3 "$root/bench.py" --no-cpu-baseline --steps 2 --warmup 1 > "$out/pmc_fetch.log" 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o run -- python3 "$root/bench.py" --no-cpu-baseline --steps 2 --warmup 1 > "$out/pmc_write.log" 2>&1
echo "pmc write done"
# the trace CSVs are large; keep the summaries only
find "$out" -name "*_kernel_trace.csv" -delete
ls -la "$out" "$out"/prof/* | tail -20
