# which copy path / SDMA engine the runtime gives the pipeline's copies (ROCclr copy log), standalone sweep vs bench.py
export AMD_LOG_LEVEL=4 AMD_LOG_MASK=768
OHS_SWEEP=32 timeout -k 10 300 python3 tools/pcie_sweep.py 2> gpurun_out/log_sweep.txt | grep chunk_blocks
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2> gpurun_out/log_bench.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['pcie_inclusive'])"
for f in sweep bench; do echo "== $f"; wc -l gpurun_out/log_$f.txt; grep -i "copy" gpurun_out/log_$f.txt | sed -e 's/0x[0-9a-f]*/X/g' -e 's/[0-9][0-9][0-9][0-9]*/N/g' | cut -c1-160 | sort | uniq -c | sort -rn | head -12; grep "HSA Copy" gpurun_out/log_$f.txt | tail -4 | cut -c1-250; done
rm -f gpurun_out/log_sweep.txt gpurun_out/log_bench.txt
