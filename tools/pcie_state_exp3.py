#!/usr/bin/env python3
"""Follow-up to pcie_state_exp2.py: is it any hipFree or only a large one, and do plain whole-tensor copies in both
directions slow down too?"""
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
bp = ohs.BatchProcessor(S, num_bands=10)
irs = synth.hrir_set(512)
for p in range(4):
    bp.set_ir(p, irs[p])
bp.set_eq_enabled(False)
x = synth.white_noise_torch(0, S, frames, dev)
hx = x.cpu().pin_memory()
hy = torch.empty_like(hx).pin_memory()
d1, d2 = torch.empty_like(x), torch.empty_like(x)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def leg(what):
    bp.process_host(hx, out=hy)
    t0 = time.perf_counter()
    for _ in range(3):
        bp.process_host(hx, out=hy)
    t = (time.perf_counter() - t0) / 3 * 1e3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        with torch.cuda.stream(s1):
            d1.copy_(hx, non_blocking=True)
        with torch.cuda.stream(s2):
            hy.copy_(d2, non_blocking=True)
        torch.cuda.synchronize()
    r = hx.numel() * 4 / ((time.perf_counter() - t0) / 3) / 1e9
    print(f"{what}: pipeline {t:.2f} ms; plain H2D || D2H {r:.1f} GB/s each way", flush=True)


leg("fresh")
for mib in (1, 16, 64, 256, 1024):
    t = torch.empty(mib << 20, dtype=torch.uint8, device=dev)
    del t
    torch.cuda.empty_cache()
    leg(f"after hipFree of {mib} MiB")
