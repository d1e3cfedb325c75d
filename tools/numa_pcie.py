#!/usr/bin/env python3
"""Does the NUMA node of the pinned host buffers decide the host pipeline's rate?  For each NUMA node of the box:
restrict this process to the node's CPUs (first-touch then places the pinned pages there), allocate the headline
workload's host buffers, time ohs_batch_process_host.  Prints the GPU's own node beside it."""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpulist(s):
    out = set()
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-"); out |= set(range(int(a), int(b) + 1))
        elif part:
            out.add(int(part))
    return out


def main():
    nodes = {}
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(d + "/cpulist").read())
    allowed = os.sched_getaffinity(0)
    print("nodes:", {n: len(c) for n, c in nodes.items()}, "allowed cpus:", len(allowed), flush=True)
    for f in glob.glob("/sys/class/drm/card*/device/numa_node"):
        print(f, open(f).read().strip())
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
    src = synth.white_noise_torch(0, S, frames, torch.device("cuda:0")).cpu()
    for n, cpus in nodes.items():
        use = cpus & allowed
        if not use:
            print(f"node {n}: no allowed cpu"); continue
        os.sched_setaffinity(0, use)
        time.sleep(0.05)
        hx = src.clone().pin_memory()
        hy = torch.empty_like(hx).pin_memory()
        hy.zero_()
        bp.reset()
        bp.process_host(hx, out=hy)
        t0 = time.perf_counter()
        for _ in range(3):
            bp.process_host(hx, out=hy)
        dt = (time.perf_counter() - t0) / 3
        print(f"buffers allocated from node {n}: {dt * 1e3:.2f} ms  {S * frames * 8 / dt / 1e9:.1f} GB/s each way", flush=True)
        del hx, hy
    os.sched_setaffinity(0, allowed)


if __name__ == "__main__":
    main()
