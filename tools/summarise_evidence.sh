#!/bin/bash
# After `bash tools/collect_evidence.sh <tag>` on the GPU box: turn gpurun_out/<tag>/ into the tracked summaries
# under profiles/ (<tag>_bench.json, _other_configs.jsonl, _kernel_stats_<workload>.csv, _pmc*.json, _counters.json).
set -e
tag=$1
o=gpurun_out/$tag
cp $o/bench.json profiles/${tag}_bench.json
cp $o/other_configs.jsonl profiles/${tag}_other_configs.jsonl
for f in rt_latency rt_latency_long rt_soak ubench_lds ubench_pcie; do [ -s $o/$f.txt ] && cp $o/$f.txt profiles/${tag}_$f.txt; done
for wl in headline c3noeq c4; do
    # kernel statistics of our own kernels only (the torch elementwise kernels of the synthetic-data generator are not ours)
    head -1 $o/kt_$wl/run_kernel_stats.csv > profiles/${tag}_kernel_stats_$wl.csv
    grep 'ohs::' $o/kt_$wl/run_kernel_stats.csv >> profiles/${tag}_kernel_stats_$wl.csv
    python3 tools/pmc_table.py ${tag}_hbm_$wl $o/pmc_fetch_$wl/run_counter_collection.csv $o/pmc_write_$wl/run_counter_collection.csv > /dev/null
done
cp profiles/${tag}_kernel_stats_headline.csv profiles/${tag}_kernel_stats.csv
python3 tools/pmc_summary.py $o/pmc_fetch_headline/run_counter_collection.csv $o/pmc_write_headline/run_counter_collection.csv $tag > /dev/null
for wl in headline c3noeq; do
    python3 tools/pmc_table.py ${tag}_sq_$wl $o/sq_a_$wl/run_counter_collection.csv $o/sq_b_$wl/run_counter_collection.csv > /dev/null
done
ls -la profiles/${tag}_*
