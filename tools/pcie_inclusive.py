#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload (never `value` in bench.py): pinned host buffers ->
H2D -> EQ + convolution + gain -> D2H, per step, with and without double buffering on two streams."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402

S, frames, taps = 256, 480256, 512
dev = torch.device("cuda:0")
bp = ohs.BatchProcessor(S, num_bands=10)
irs = synth.hrir_set(taps)
for p in range(4):
    bp.set_ir(p, irs[p])
for i, b in enumerate(synth.eq_table()):
    bp.update_band_coeffs(i, synth.FS, b)
bp.set_eq_enabled(True)
hx = synth.white_noise_torch(0, S, frames, dev).cpu().pin_memory()
hy = torch.empty_like(hx).pin_memory()
dx = torch.empty((S, 2, frames), device=dev)
dy = torch.empty_like(dx)


def step():
    dx.copy_(hx, non_blocking=True)
    bp.process(dx, out=dy)
    hy.copy_(dy, non_blocking=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
gb = 2 * S * 2 * frames * 4 / 1e9
print(json.dumps({"config": "c3_pcie_inclusive_serial", "ms_per_step": round(dt * 1e3, 2),
                  "Msamples_per_s": round(S * frames / dt / 1e6, 1), "host_device_GB_per_step": round(gb, 3),
                  "effective_GBps": round(gb / dt, 1)}))
