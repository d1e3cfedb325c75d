"""CPU suite (no GPU): the oracle against the committed golden vectors (tests/golden, made by
tools/make_golden.py from the f64 ground truth) and against independent numpy computations."""
import glob
import os

import numpy as np
import pytest

from tests.util import assert_parity, rms_err

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "conv_*.npz"))))
def test_oracle_engine_matches_golden_f64(oracle, path):
    g = np.load(path)
    e = oracle.ConvolutionEngine()
    for p in range(4):
        e.set_ir(p, g["irs"][p])
    x = g["x"]
    outs = []
    for c in range(x.shape[1] // 1024):
        outs.append(np.stack(e.process_block(x[0, c * 1024:(c + 1) * 1024], x[1, c * 1024:(c + 1) * 1024])))
    y = np.concatenate(outs, axis=1)
    assert_parity(y, g["y_f64"], os.path.basename(path))


def test_oracle_eq_matches_golden_bits(oracle):
    g = np.load(os.path.join(GOLD, "eq_11bands_3000.npz"))
    eq = oracle.StereoParametricEQ(len(g["enabled"]), 48000.0)
    for i in range(len(g["enabled"])):
        eq.set_band_coeffs(i, g["coeffs"][i], bool(g["enabled"][i]))
    l, r = g["x"][0].copy(), g["x"][1].copy()
    eq.process_block(l, r)
    assert np.array_equal(np.stack([l, r]).view(np.uint32), g["y_f32"].view(np.uint32))
    # the f32 recurrence's own rounding noise vs exact arithmetic (why the GPU must be bit-exact)
    print("f32 DF2T vs f64:", rms_err(g["y_f32"], g["y_f64"]))


def test_oracle_fft_matches_numpy(oracle):
    rng = np.random.default_rng(3)
    z = (rng.standard_normal(1024) + 1j * rng.standard_normal(1024)).astype(np.complex64)
    Z = oracle.fft1024(z)
    ref = np.fft.fft(z.astype(np.complex128))
    assert np.linalg.norm(Z - ref) / np.linalg.norm(ref) < 3e-7
    back = oracle.fft1024(Z, inverse=True) / 1024
    assert np.linalg.norm(back - z) / np.linalg.norm(z) < 5e-7


def test_oracle_direct_conv_matches_numpy(oracle):
    rng = np.random.default_rng(4)
    x = rng.uniform(-1, 1, 3000).astype(np.float32)
    h = rng.standard_normal(257).astype(np.float32)
    y = oracle.direct_conv_f64(x, h)
    ref = np.convolve(x.astype(np.float64), h.astype(np.float64))[: x.size]
    assert np.allclose(y, ref, rtol=0, atol=1e-12)


def test_oracle_biquad_coefficients_against_rbj_cookbook(oracle):
    """Independent float64 evaluation of the RBJ cookbook forms (the published algorithm the
    biquad crate implements); the f32 restatement must agree to f32 rounding."""
    fs = 48000.0
    for t, fc, q, g in [(oracle.PEAK, 1000.0, 1.41, 3.0), (oracle.LOWSHELF, 105.0, 0.7, 4.0),
                        (oracle.HIGHSHELF, 10000.0, 0.7, -3.0), (oracle.LOWPASS, 1000.0, 0.707, 0.0),
                        (oracle.HIGHPASS, 80.0, 0.5, 0.0), (oracle.BANDPASS, 440.0, 2.0, 0.0),
                        (oracle.NOTCH, 60.0, 5.0, 0.0), (oracle.ALLPASS, 3000.0, 1.0, 0.0)]:
        w = 2 * np.pi * fc / fs
        sn, cs = np.sin(w), np.cos(w)
        al = sn / (2 * q)
        A = 10 ** (g / 40)
        if t == oracle.PEAK:
            b = [1 + al * A, -2 * cs, 1 - al * A]; a = [1 + al / A, -2 * cs, 1 - al / A]
        elif t == oracle.LOWSHELF:
            s2 = 2 * np.sqrt(A) * al
            b = [A * ((A + 1) - (A - 1) * cs + s2), 2 * A * ((A - 1) - (A + 1) * cs), A * ((A + 1) - (A - 1) * cs - s2)]
            a = [(A + 1) + (A - 1) * cs + s2, -2 * ((A - 1) + (A + 1) * cs), (A + 1) + (A - 1) * cs - s2]
        elif t == oracle.HIGHSHELF:
            s2 = 2 * np.sqrt(A) * al
            b = [A * ((A + 1) + (A - 1) * cs + s2), -2 * A * ((A - 1) + (A + 1) * cs), A * ((A + 1) + (A - 1) * cs - s2)]
            a = [(A + 1) - (A - 1) * cs + s2, 2 * ((A - 1) - (A + 1) * cs), (A + 1) - (A - 1) * cs - s2]
        elif t == oracle.LOWPASS:
            b = [(1 - cs) / 2, 1 - cs, (1 - cs) / 2]; a = [1 + al, -2 * cs, 1 - al]
        elif t == oracle.HIGHPASS:
            b = [(1 + cs) / 2, -(1 + cs), (1 + cs) / 2]; a = [1 + al, -2 * cs, 1 - al]
        elif t == oracle.BANDPASS:
            b = [sn / 2, 0, -sn / 2]; a = [1 + al, -2 * cs, 1 - al]
        elif t == oracle.NOTCH:
            b = [1, -2 * cs, 1]; a = [1 + al, -2 * cs, 1 - al]
        else:
            b = [1 - al, -2 * cs, 1 + al]; a = [1 + al, -2 * cs, 1 - al]
        ref = np.array([b[0] / a[0], b[1] / a[0], b[2] / a[0], a[1] / a[0], a[2] / a[0]])
        got = oracle.biquad_coeffs(t, fs, fc, q, g)
        # low-frequency shelves/peaks lose digits in f32 through 1 - cos(w); 2e-4 relative is ample
        assert np.allclose(got, ref, rtol=2e-4, atol=1e-6), (t, got, ref)
    with pytest.raises(ValueError):
        oracle.biquad_coeffs(oracle.PEAK, fs, 30000.0, 1.0, 0.0)
    with pytest.raises(ValueError):
        oracle.biquad_coeffs(oracle.PEAK, fs, 1000.0, -0.5, 0.0)


def test_oracle_semantics_fifo_silence_and_clone(oracle):
    """convolution.rs:163-181: fewer than n outputs buffered => zeros, FIFO keeps its content."""
    e = oracle.ConvolutionEngine()
    e.set_ir(oracle.LSL, [1.0]); e.set_ir(oracle.RSR, [1.0])
    x = np.arange(1, 257, dtype=np.float32)
    l, r = e.process_block(x, x)                 # 256 buffered, nothing out yet
    assert not np.any(l) and not np.any(r)
    c = e.clone()
    l, r = e.process_block(x, x)                 # 512 in -> one block -> 512 out >= 256
    assert np.allclose(l, x, atol=1e-4)          # first 256 outputs = first 256 inputs
    l2, r2 = c.process_block(x, x)               # the clone carries the FIFO too
    assert np.array_equal(l, l2) and np.array_equal(r, r2)
    assert e.num_partitions(oracle.LSR) == 1     # default path: one zero partition (:46-63)


def test_oracle_set_ir_resets_only_that_path(oracle):
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (2, 2048)).astype(np.float32)
    h = [rng.standard_normal(600).astype(np.float32) * 0.05 for _ in range(4)]
    e = oracle.ConvolutionEngine()
    for p in range(4):
        e.set_ir(p, h[p])
    e.process_block(x[0, :1024], x[1, :1024])
    e.set_ir(oracle.LSL, h[0])                   # same IR again: only Lsl forgets its past
    l, r = e.process_block(x[0, 1024:], x[1, 1024:])
    # right ear (lsr + rsr) is unaffected: equals an engine that never reset
    e2 = oracle.ConvolutionEngine()
    for p in range(4):
        e2.set_ir(p, h[p])
    e2.process_block(x[0, :1024], x[1, :1024])
    l2, r2 = e2.process_block(x[0, 1024:], x[1, 1024:])
    assert np.array_equal(r, r2) and not np.array_equal(l, l2)
    # left ear = rsl with history + lsl restarted from silence
    f_rsl = oracle.direct_conv_f64(x[1], h[2])[1024:]
    f_lsl = oracle.direct_conv_f64(x[0, 1024:], h[0])
    assert_parity(l, f_rsl + f_lsl, "lsl restarted")
