#!/usr/bin/env python3
"""BASELINE config 4 (64 streams x 16 384 taps x 131 072 frames, EQ off) under library variants: per-step time by HIP
events after a warm-up, interleaved over the libraries given.   python tools/ab_c4.py libA.so libB.so ...  (run as
separate processes per library: a process loads one libohs_hip)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
import open_headstage_amd as ohs
from open_headstage_amd import synth
S, frames, taps = 64, 131072, 16384
bp = ohs.BatchProcessor(S, num_bands=10)
for p, h in enumerate(synth.hrir_set(taps)):
    bp.set_ir(p, h)
bp.set_eq_enabled(False)
x = synth.white_noise_torch(0, S, frames, torch.device("cuda:0")); y = torch.empty_like(x)
for _ in range(60):
    bp.process(x, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for rep in range(5):
    e0.record()
    for _ in range(40):
        bp.process(x, out=y)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 40)
print(" ".join("%%.4f" %% t for t in ts), "ms per step; finite", bool(torch.isfinite(y).all()), "checksum %%.6e" %% float(y.double().abs().sum()))
''' % ROOT

for rep in range(2):
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, OHS_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        print(os.path.basename(lib), (r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)
