#!/usr/bin/env python3
"""A/B of the two one-partition plans (ohs_batch_set_conv_plan): block 512 / FFT 1024 (k_conv_p1) against hop 1536 /
FFT 2048 (k_conv_p1_os).  Interleaved timing with HIP events, EQ off (BASELINE config 3 without EQ) and on, at several
stream counts.  python tools/ab_conv_plan.py [streams ...] > profiles/r04_conv_plan_ab.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402

FRAMES = 480256
dev = torch.device("cuda:0")
streams = [int(a) for a in sys.argv[1:]] or [256, 1024, 2048]


def make(S, plan, eq):
    bp = ohs.BatchProcessor(S, num_bands=10)
    irs = synth.hrir_set(512)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(synth.eq_table()):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(eq)
    bp.set_gain(0.8912509)
    bp.set_conv_plan(plan)
    return bp


def timed(bp, x, y, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        bp.process(x, out=y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for S in streams:
    x = synth.white_noise_torch(0, S, FRAMES, dev)
    y = torch.empty_like(x)
    for eq in (False, True):
        bps = {1: make(S, 1, eq), 2: make(S, 2, eq)}
        for plan in (1, 2):
            for _ in range(30 if not eq else 3):
                bps[plan].process(x, out=y)
        torch.cuda.synchronize()
        rows = {1: [], 2: []}
        for rep in range(4):
            for plan in (1, 2):
                rows[plan].append(timed(bps[plan], x, y, 12 if not eq else 5))
        nb = FRAMES // 512
        line = []
        for plan in (1, 2):
            best = min(rows[plan])
            _, cb = bps[plan].kernel_bytes(nb)
            line.append(f"plan {plan}: " + " ".join(f"{t:.4f}" for t in rows[plan]) + f" ms (best {best:.4f}"
                        + (f" = {S * FRAMES * 16 / best / 1e6 / 8000 * 100:.1f} % of HBM peak by 16 B/frame" if not eq else "") + ")")
        print(f"{S} streams x {FRAMES} frames, EQ {'on' if eq else 'off'}:  " + "  |  ".join(line)
              + f"  ->  {min(rows[2]) / min(rows[1]):.3f} x", flush=True)
        del bps
    del x, y
    torch.cuda.empty_cache()
