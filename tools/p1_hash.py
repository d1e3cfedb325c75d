#!/usr/bin/env python3
"""SHA-256 of the P = 1 convolution's output and state on fixed seeded inputs, several shapes: two builds of the library
whose arithmetic is meant to be identical (e.g. the paired against the natural bin layout of k_conv_p1,
OHS_BUILD_TAG=natural OHS_EXTRA_DEFS=-DOHS_P1_NATURAL) must print the same lines.
    python tools/p1_hash.py; OHS_LIB=open_headstage_amd/libohs_hip_natural.so python tools/p1_hash.py"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
for S, blocks, calls in ((1, 1, 3), (3, 37, 2), (5, 300, 2), (64, 938, 1), (256, 938, 2), (700, 64, 1)):
    bp = ohs.BatchProcessor(S, num_bands=10)
    irs = synth.hrir_set(512)
    for p in range(4):
        bp.set_ir(p, irs[p])
    bp.set_eq_enabled(False)
    bp.set_gain(0.7)
    bp.set_conv_plan(1)     # the block-512 plan: the bits this file records
    h = hashlib.sha256()
    for c in range(calls):
        x = synth.white_noise_torch(10 * S + c, S, blocks * 512, dev)
        y = torch.empty_like(x)
        bp.process(x, out=y)
        torch.cuda.synchronize()
        h.update(y.cpu().numpy().tobytes())
    print(S, blocks, calls, h.hexdigest(), flush=True)
    del bp
