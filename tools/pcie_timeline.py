#!/usr/bin/env python3
"""Three ohs_batch_process_host calls of the headline workload, to be run under
   rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl -o tl --output-format csv -- python3 tools/pcie_timeline.py
and analysed with tools/pcie_timeline.py --analyse gpurun_out/tl: per call, busy time of each copy direction and of
the kernels, and the gaps between consecutive copies of one direction."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    S, frames = 256, 480256
    bp = ohs.BatchProcessor(S, num_bands=10)
    irs = synth.hrir_set(512)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(synth.eq_table()):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    hx = synth.white_noise_torch(0, S, frames, torch.device("cuda:0")).cpu().pin_memory()
    hy = torch.empty_like(hx).pin_memory()
    cb = int(os.environ.get("OHS_TL_CHUNK_BLOCKS", "0"))
    for _ in range(3):
        bp.process_host(hx, out=hy, chunk_blocks=cb)


def analyse(d):
    copies, kernels = [], []
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kernels.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    copies.sort()
    big = [c for c in copies if c[1] - c[0] > 200_000]      # the chunk copies (> 0.2 ms)
    if not big:
        print("no chunk copies found"); return
    # the last call = the last third of the big copies
    n = len(big) // 3
    call = big[-n:]
    t0 = min(c[0] for c in call); t1 = max(c[1] for c in call)
    print(f"last call: {n} chunk copies, span {(t1 - t0) / 1e6:.2f} ms")
    for direction in sorted({c[2] for c in call}):
        cs = [c for c in call if c[2] == direction]
        busy = sum(c[1] - c[0] for c in cs)
        gaps = [(cs[i + 1][0] - cs[i][1]) / 1e3 for i in range(len(cs) - 1)]
        print(f"  {direction:>16}: {len(cs)} copies, busy {busy / 1e6:.2f} ms, each {[round((c[1] - c[0]) / 1e6, 2) for c in cs]} ms")
        print(f"  {'':>16}  first starts at {(cs[0][0] - t0) / 1e6:.2f} ms, last ends at {(cs[-1][1] - t0) / 1e6:.2f} ms, gaps [us] {[round(g) for g in gaps]}")
    ks = [k for k in kernels if t0 <= k[0] <= t1]
    print(f"  kernels in the span: {len(ks)}, busy {sum(k[1] - k[0] for k in ks) / 1e6:.2f} ms")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        analyse(sys.argv[2])
    else:
        run()
