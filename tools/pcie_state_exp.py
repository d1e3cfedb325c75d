#!/usr/bin/env python3
"""What in bench.py's life before its PCIe leg slows the host pipeline from 21.7 to 25.6 ms?  One process, the
pipeline timed after each stage: fresh / after compute steps / after the saturation churn / with new pinned buffers."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402

S, frames = 256, 480256
dev = torch.device("cuda:0")


def make(S_):
    bp = ohs.BatchProcessor(S_, num_bands=10)
    irs = synth.hrir_set(512)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(synth.eq_table()):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    return bp


def leg(bp, hx, hy, what):
    bp.process_host(hx, out=hy)
    t0 = time.perf_counter()
    for _ in range(3):
        bp.process_host(hx, out=hy)
    print(f"{what}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)


bp = make(S)
x = synth.white_noise_torch(0, S, frames, dev)
y = torch.empty_like(x)
hx = x.cpu().pin_memory()
hy = torch.empty_like(hx).pin_memory()
leg(bp, hx, hy, "fresh")
for _ in range(13):
    bp.process(x, out=y)
torch.cuda.synchronize()
leg(bp, hx, hy, "after 13 device steps")
if "profiling" in sys.argv:
    bp.set_profiling(True)
    for _ in range(10):
        bp.process(x, out=y)
    torch.cuda.synchronize()
    bp.profile_read(); bp.set_profiling(False)
    leg(bp, hx, hy, "after 10 profiled steps")
del x, y
torch.cuda.empty_cache()
leg(bp, hx, hy, "after freeing the device tensors")
for S2 in (1024, 2048):
    bp2 = make(S2)
    x2 = synth.white_noise_torch(0, S2, frames, dev)
    y2 = torch.empty_like(x2)
    for _ in range(4):
        bp2.process(x2, out=y2)
    torch.cuda.synchronize()
    del bp2, x2, y2
    torch.cuda.empty_cache()
    leg(bp, hx, hy, f"after the {S2}-stream run")
hx2 = synth.white_noise_torch(0, S, frames, dev).cpu().pin_memory()
hy2 = torch.empty_like(hx2).pin_memory()
leg(bp, hx2, hy2, "new pinned buffers")
bp.reset()
leg(bp, hx2, hy2, "after reset")
leg(bp, hx, hy, "old pinned buffers again")
