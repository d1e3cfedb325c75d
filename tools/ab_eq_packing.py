#!/usr/bin/env python3
"""What two EQ chains per DPP row (packed halves of v_pk_* operations) could be worth: the measurement that bounds it.
A packed wave runs two chains in 12-13 instructions per step instead of 6.5 per chain, i.e. it replaces TWO waves of the
ring kernel on one SIMD by ONE wave of ~1.85-2.0 x the step time.  So the gain is (time of two ring waves sharing a SIMD)
/ (time of one packed wave) - and the numerator is measurable today: the EQ kernel's time per step at 2048 streams (one
wave per SIMD), at 4096 (two) and at 8192 (four).  python tools/ab_eq_packing.py > profiles/r04_eq_two_chains_per_row.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
FRAMES = int(os.environ.get("FRAMES", "120320"))        # 235 blocks: a quarter of the headline step, EQ time is linear in it
res = {}
for S in (1024, 2048, 4096, 8192):
    bp = ohs.BatchProcessor(S, num_bands=10)
    irs = synth.hrir_set(512)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(synth.eq_table()):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    x = synth.white_noise_torch(0, S, FRAMES, dev)
    for _ in range(3):
        bp.process(x, out=x)
    torch.cuda.synchronize()
    bp.set_profiling(True)
    for _ in range(5):
        bp.process(x, out=x)
    torch.cuda.synchronize()
    eq_ms, conv_ms, n, eq_l, conv_l = bp.profile_read()
    res[S] = (eq_ms / n, conv_ms / n, eq_l // n)
    waves_per_simd = S * 2 / 4 / 1024
    print(f"{S:5d} streams x {FRAMES} frames: EQ {eq_ms / n:7.3f} ms per step in {eq_l // n} launch(es) "
          f"({waves_per_simd:.1f} ring waves per SIMD; {eq_ms / n * 1e6 / FRAMES:.2f} ns per sample per chain-group), "
          f"convolution {conv_ms / n:6.3f} ms", flush=True)
    del bp, x
    torch.cuda.empty_cache()
one, two, four = res[2048][0], res[4096][0], res[8192][0]
print(f"two ring waves on a SIMD take {two / one:.3f} x one, four take {four / one:.3f} x")
print(f"a packed wave (two chains, 12 instructions per step against 6.5) takes >= {12 / 6.5:.3f} x one ring wave at the same "
      f"clock: against two ring waves it is worth at most {(two / one) / (12 / 6.5):.3f} x (13 instructions: "
      f"{(two / one) / 2.0:.3f} x) -- and at 2048 streams no SIMD hosts two ring waves to begin with.")
