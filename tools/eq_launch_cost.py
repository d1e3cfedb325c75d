#!/usr/bin/env python3
"""Fixed cost of one k_eq_ring launch: EQ span (HIP events around the launch, library profiling) of a 256-stream batch
call for several block counts, one EQ launch per call (OHS_NO_OVERLAP=1 is set here); least-squares a + b * n."""
import os
import sys

os.environ["OHS_NO_OVERLAP"] = "1"
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402

S = 256
dev = torch.device("cuda:0")
bp = ohs.BatchProcessor(S, num_bands=10)
irs = synth.hrir_set(512)
for p in range(4):
    bp.set_ir(p, irs[p])
for i, b in enumerate(synth.eq_table()):
    bp.update_band_coeffs(i, synth.FS, b)
bp.set_eq_enabled(True)
ns, ts = [], []
for nb in (1, 2, 4, 8, 19, 56, 160, 234, 469, 938):
    x = synth.white_noise_torch(0, S, nb * 512, dev)
    y = torch.empty_like(x)
    for _ in range(3):
        bp.process(x, out=y)
    torch.cuda.synchronize()
    bp.set_profiling(True)
    for _ in range(8):
        bp.process(x, out=y)
    torch.cuda.synchronize()
    eq_ms, conv_ms, n, _, _ = bp.profile_read()
    bp.set_profiling(False)
    print(f"blocks {nb:4d}: EQ launch {eq_ms / n * 1e3:9.1f} us  ({eq_ms / n * 1e6 / (nb * 512):7.2f} ns per sample)", flush=True)
    ns.append(nb * 512); ts.append(eq_ms / n * 1e3)
A = np.vstack([np.ones(len(ns)), np.array(ns, float)]).T
(a, b), *_ = np.linalg.lstsq(A, np.array(ts), rcond=None)
print(f"fit: {a:.1f} us per launch + {b * 1e3:.3f} ns per sample")
