#!/usr/bin/env python3
"""In-process A/B timing of k_conv_p1 launch knobs (OHS_P1_TARGET_WAVES, OHS_P1_STAGGER) on BASELINE config 3
without EQ: one batch object, the settings cycled round-robin for several rounds, HIP-event time per step.
Boxes and runs differ by a few per cent; only comparisons inside one process are meaningful.
usage: python tools/tune_p1.py [streams]"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = 480256
dev = torch.device("cuda:0")
bp = ohs.BatchProcessor(S, num_bands=10)
irs = synth.hrir_set(512)
for p in range(4):
    bp.set_ir(p, irs[p])
bp.set_eq_enabled(False)
x = synth.white_noise_torch(0, S, frames, dev)
y = torch.empty_like(x)
targets = [int(t) for t in os.environ.get("TUNE_P1_TARGETS", "3840,4096,7680,8192,11520,12288").split(",")]
settings = [dict(OHS_P1_TARGET_WAVES=str(tw), OHS_P1_STAGGER=str(sg)) for tw, sg in itertools.product(targets, (0,))]
res = {i: [] for i in range(len(settings))}
for rnd in range(6):
    for i, st in enumerate(settings):
        os.environ.update(st)
        bp.process(x, out=y)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(8):
            bp.process(x, out=y)
        b.record()
        torch.cuda.synchronize()
        res[i].append(a.elapsed_time(b) / 8)
for i, st in enumerate(settings):
    v = np.array(res[i][1:])
    print(st, f"median {np.median(v):.4f} ms  min {v.min():.4f}  max {v.max():.4f}", flush=True)
