#!/bin/bash
# Round evidence on the GPU box:  gpurun --timeout 1100 -- 'bash tools/collect_evidence.sh r02_v1'
# Writes under gpurun_out/<tag>/ ; afterwards, locally:  bash tools/summarise_evidence.sh <tag>   (copies the
# summaries into profiles/).  PMC passes are separate runs, without any trace domain (pool rule), the program
# itself after `--`.
set -e -o pipefail
tag=${1:-evidence}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --steps 10 --warmup 3 > "$out/bench.json" 2> "$out/bench.err"
echo "bench done"; cat "$out/bench.json"
python3 tools/bench_configs.py > "$out/other_configs.jsonl" 2> "$out/other_configs.err"
# native latency harness: the JSON lines join the other configs, the whole output (with the device-side phases) is kept too
timeout -k 10 120 tools/bin/rt_latency > "$out/rt_latency.txt" 2>> "$out/other_configs.err" || true
grep -v '^#' "$out/rt_latency.txt" >> "$out/other_configs.jsonl" || true
timeout -k 10 120 tools/bin/rt_latency long > "$out/rt_latency_long.txt" 2>> "$out/other_configs.err" || true
grep -v '^#' "$out/rt_latency_long.txt" >> "$out/other_configs.jsonl" || true
timeout -k 10 120 tools/bin/rt_latency soak 300000 > "$out/rt_soak.txt" 2>> "$out/other_configs.err" || true
# micro-benchmarks the design decisions quote
timeout -k 10 120 tools/bin/ubench_lds > "$out/ubench_lds.txt" 2>&1 || true
timeout -k 10 120 tools/bin/ubench_pcie > "$out/ubench_pcie.txt" 2>&1 || true
echo "other configs done"
cd /tmp
prof() {   # prof <name> <rocprofv3 args...> -- <program...>
    local name=$1; shift
    rocprofv3 "$@" > "$out/$name.log" 2>&1
    echo "$name done"
}
BENCH=(python3 "$root/bench.py" --no-cpu-baseline --no-extras)
CFG=(python3 "$root/tools/bench_configs.py")
prof kt_headline --kernel-trace --stats --output-format csv -d "$out/kt_headline" -o run -- "${BENCH[@]}" --steps 5 --warmup 2
prof kt_c3noeq   --kernel-trace --stats --output-format csv -d "$out/kt_c3noeq" -o run -- "${CFG[@]}" c3noeq
prof kt_c4       --kernel-trace --stats --output-format csv -d "$out/kt_c4" -o run -- "${CFG[@]}" c4
for wl in headline c3noeq c4; do
    case $wl in headline) P=("${BENCH[@]}" --steps 2 --warmup 1) ;; c3noeq) P=("${CFG[@]}" c3noeq) ;; c4) P=("${CFG[@]}" c4) ;; esac
    prof pmc_fetch_$wl --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch_$wl" -o run -- "${P[@]}"
    prof pmc_write_$wl --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write_$wl" -o run -- "${P[@]}"
done
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
for wl in headline c3noeq; do
    case $wl in headline) P=("${BENCH[@]}" --steps 2 --warmup 1) ;; c3noeq) P=("${CFG[@]}" c3noeq) ;; esac
    prof sq_a_$wl --pmc $A --output-format csv -d "$out/sq_a_$wl" -o run -- "${P[@]}"
    prof sq_b_$wl --pmc $B --output-format csv -d "$out/sq_b_$wl" -o run -- "${P[@]}"
done
# the trace CSVs are large; keep the summaries only
find "$out" -name "*_kernel_trace.csv" -delete
du -sh "$out"; ls "$out"
