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


def eq_ftz_case(name):
    """The EQ in the three denormal modes (ohs_*_set_flush_denormals; tests/test_gpu_ftz.py): states decaying through the
    denormal range, excited again later, plus denormal INPUT samples -- the oracle's f32 DF2T run under MXCSR = IEEE /
    FTZ / FTZ | DAZ on the authoring machine.  Committed so that the GPU test does not depend on how the test box's CPU
    (or its libc) treats MXCSR."""
    from open_headstage_amd import BandConfig, FilterType, biquad_coefficients
    bands = [BandConfig(FilterType.LowPass, 900.0, 0.6, 0.0, True), BandConfig(FilterType.Peak, 300.0, 2.0, 6.0, True),
             BandConfig(FilterType.HighShelf, 6000.0, 0.7, -4.0, True)]
    coeffs = np.stack([biquad_coefficients(b.filter_type, 48000.0, b.center_freq, b.q, b.gain_db) for b in bands])
    n = 6000
    x = np.zeros((2, n), np.float32)
    x[0, 0], x[1, 3] = 3e-33, -7e-34
    x[0, 3000:3010] = 1e-36
    xin = np.zeros((2, 2000), np.float32)           # denormal input samples
    xin[0, ::7] = np.float32(1.1e-38)
    xin[1, ::5] = np.float32(-9e-39)
    coeffs_in = coeffs.copy()
    coeffs_in[1] = biquad_coefficients(FilterType.Peak, 48000.0, 300.0, 2.0, 12.0)
    out = {"coeffs": coeffs, "coeffs_in": coeffs_in, "x": x, "x_in": xin, "call": np.int64(1000)}
    for tag, xx, cc in (("y", x, coeffs), ("y_in", xin, coeffs_in)):
        for mode in (0, 1, 2):
            eq = orc.StereoParametricEQ(len(bands), 48000.0)
            for i in range(len(bands)):
                eq.set_band_coeffs(i, cc[i], True)
            l, r = xx[0].copy(), xx[1].copy()
            with orc.flush_denormals(mode):
                for pos in range(0, xx.shape[1], 1000):         # state carried across calls, as in the test
                    a, b = l[pos:pos + 1000].copy(), r[pos:pos + 1000].copy()
                    eq.process_block(a, b)
                    l[pos:pos + 1000], r[pos:pos + 1000] = a, b
            out[f"{tag}_mode{mode}"] = np.stack([l, r])
    # sanity of the generator itself: the modes really differ, IEEE really walks through denormals, FTZ never emits one
    den = lambda y: int(((np.abs(y) > 0) & (np.abs(y) < np.float32(1.1754944e-38))).sum())      # noqa: E731
    assert den(out["y_mode0"]) > 100 and den(out["y_mode1"]) == 0 and den(out["y_mode2"]) == 0
    assert not np.array_equal(out["y_mode0"], out["y_mode1"]) and np.count_nonzero(out["y_in_mode2"]) == 0
    np.savez_compressed(os.path.join(OUT, name), **out)


if __name__ == "__main__":
    eq_ftz_case("eq_ftz_modes.npz")
    conv_case("conv_c1_128taps_1024.npz", 128, 1024, 100)      # BASELINE configs[0]
    conv_case("conv_c2_512taps_4096.npz", 512, 4096, 101)      # configs[1] shape
    conv_case("conv_p3_1300taps_4096.npz", 1300, 4096, 102)    # three partitions
    eq_case("eq_11bands_3000.npz", 3000, 103)
    print("wrote", sorted(os.listdir(OUT)))
