#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- small golden input/output vectors for the hot path.

PROVENANCE (read before trusting): the reference (Rust + crates.io deps + libmysofa) cannot be
built or run in the authoring container, so these vectors are NOT outputs of the reference.
They are produced by this repository's own ground truth in oracle/ohs_oracle.c:
  * convolution : f64 direct time-domain convolution y[n] = sum_k h[k] x[n-k] per path, ear sums
                  out_l = lsl + rsl, out_r = lsr + rsr (routing of convolution.rs:193-231);
  * EQ          : the f32 DF2T recurrence of parametric_eq.rs:116-122 (bit-exact target) and its
                  f64 twin, both with the f32 coefficients stored in the fixture.
Inputs come from the portable splitmix64 generators (open_headstage_amd/synth.py).
Run from the repo root:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_headstage_amd import synth  # noqa: E402
from oracle import ohs_oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def conv_case(name, taps, frames, sid):
    irs = synth.hrir_set(taps)
    x = synth.white_noise([sid], frames)[0]
    out_l, out_r = orc.binaural_f64(x[0], x[1], irs)
    np.savez_compressed(os.path.join(OUT, name), taps=taps, x=x, irs=np.stack(irs),
                        y_f64=np.stack([out_l, out_r]))


def eq_case(name, frames, sid):
    bands = synth.eq_table() + [synth.eq_stress_band()]
    coeffs = np.stack([orc.biquad_coeffs(int(b.filter_type), synth.FS, b.center_freq, b.q, b.gain_db)
                       for b in bands]).astype(np.float32)
    en = np.array([b.enabled for b in bands], np.int32)
    x = synth.white_noise([sid], frames)[0]
    eq = orc.StereoParametricEQ(len(bands), synth.FS)
    for i in range(len(bands)):
        eq.set_band_coeffs(i, coeffs[i], bool(en[i]))
    l, r = x[0].copy(), x[1].copy()
    eq.process_block(l, r)
    y64 = np.stack([orc.eq_f64(coeffs, en, x[0])[0], orc.eq_f64(coeffs, en, x[1])[0]])
    np.savez_compressed(os.path.join(OUT, name), x=x, coeffs=coeffs, enabled=en,
                        y_f32=np.stack([l, r]), y_f64=y64)


if __name__ == "__main__":
    conv_case("conv_c1_128taps_1024.npz", 128, 1024, 100)      # BASELINE configs[0]
    conv_case("conv_c2_512taps_4096.npz", 512, 4096, 101)      # configs[1] shape
    conv_case("conv_p3_1300taps_4096.npz", 1300, 4096, 102)    # three partitions
    eq_case("eq_11bands_3000.npz", 3000, 103)
    print("wrote", sorted(os.listdir(OUT)))
