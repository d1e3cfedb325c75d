#!/bin/bash
# SQ counter passes for config 4 under one plan of tools/bench_c4_plans.py:  bash tools/collect_sq_c4.sh <tag> <plan>
# afterwards: python tools/pmc_table.py <name> gpurun_out/<tag>/sq_*/run_counter_collection.csv
set -e -o pipefail
tag=${1:-sq_c4}; plan=${2:-2}
root=$(pwd); out=$root/gpurun_out/$tag
mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
P=(python3 "$root/tools/bench_c4_plans.py" --plans "$plan" --steps 6)
rocprofv3 --pmc $A --output-format csv -d "$out/sq_a" -o run -- "${P[@]}" > "$out/sq_a.log" 2>&1
rocprofv3 --pmc $B --output-format csv -d "$out/sq_b" -o run -- "${P[@]}" > "$out/sq_b.log" 2>&1
cd "$root"; python3 tools/pmc_table.py "$tag" "$out"/sq_a/run_counter_collection.csv "$out"/sq_b/run_counter_collection.csv
