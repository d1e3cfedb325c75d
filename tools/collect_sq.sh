#!/bin/bash
# SQ counter passes (VALU / LDS activity, waits, occupancy) for the kernels of one workload:
#   gpurun --timeout 900 -- 'bash tools/collect_sq.sh <tag> <workload>'
#   workload: headline | c3noeq | c4      (headline = bench.py; the others = tools/bench_configs.py)
# Two passes of 8 SQ counters each (gfx950 has 8 SQ slots per pass), PMC only -- no trace domain (pool rule),
# the program itself after `--`.  Afterwards, locally:
#   python tools/pmc_table.py <tag>_<workload> gpurun_out/<tag>/sq_<workload>_*/*_counter_collection.csv
set -e -o pipefail
tag=${1:-sq}
wl=${2:-c3noeq}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
case "$wl" in
  headline) prog=("$root/bench.py" --no-cpu-baseline --steps 2 --warmup 1) ;;
  c3noeq)   prog=("$root/tools/bench_configs.py" c3noeq) ;;
  c4)       prog=("$root/tools/bench_configs.py" c4) ;;
  *) echo "unknown workload $wl"; exit 2 ;;
esac
cd /tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
rocprofv3 --pmc $A --output-format csv -d "$out/sq_${wl}_a" -o run -- python3 "${prog[@]}" > "$out/sq_${wl}_a.log" 2>&1
echo "pass A done"
rocprofv3 --pmc $B --output-format csv -d "$out/sq_${wl}_b" -o run -- python3 "${prog[@]}" > "$out/sq_${wl}_b.log" 2>&1
echo "pass B done"
ls -la "$out"/sq_${wl}_a "$out"/sq_${wl}_b | tail -8
