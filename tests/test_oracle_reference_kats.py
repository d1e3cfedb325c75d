"""Pins the CPU oracle against the reference's OWN unit tests for the hot path.

Each test re-creates, input for input and tolerance for tolerance, one
`#[test]` of /root/reference:
  src/dsp/convolution.rs:317-347   test_identity_ir_passthrough
  src/dsp/convolution.rs:349-383   test_delay_ir
  src/dsp/convolution.rs:385-421   test_long_ir_partitioning
  src/dsp/parametric_eq.rs:218-225 test_biquad_filter_passthrough_when_disabled
  src/dsp/parametric_eq.rs:227-238 test_biquad_filter_processes_when_enabled
These are the only known-answer vectors the reference holds for this path
(SURVEY.md section 8c); the same cases run against the HIP path in
tests/test_gpu_reference_kats.py.
"""
import numpy as np

BLOCK_SIZE = 512
TOLERANCE = 1e-3  # convolution.rs:301


def _assert_approx(a, b, msg):
    assert a.shape == b.shape, msg
    assert np.all(np.abs(a - b) < TOLERANCE), msg


def test_identity_ir_passthrough(oracle):
    e = oracle.ConvolutionEngine()
    e.set_ir(oracle.LSL, [1.0])
    e.set_ir(oracle.LSR, [0.0])
    e.set_ir(oracle.RSL, [0.0])
    e.set_ir(oracle.RSR, [1.0])
    i = np.arange(BLOCK_SIZE, dtype=np.float32)
    in_l = np.sin(i * np.float32(0.1)).astype(np.float32)
    in_r = np.sin(i * np.float32(-0.1)).astype(np.float32)
    e.process_block(in_l, in_r)
    out_l, out_r = e.process_block(in_l, in_r)
    _assert_approx(out_l, in_l, "Identity passthrough L channel")
    _assert_approx(out_r, in_r, "Identity passthrough R channel")


def test_delay_ir(oracle):
    e = oracle.ConvolutionEngine()
    delay = 5
    ir = np.zeros(delay + 1, np.float32)
    ir[delay] = 1.0
    e.set_ir(oracle.LSL, ir)
    for p in (oracle.LSR, oracle.RSL, oracle.RSR):
        e.set_ir(p, [0.0])
    in_l = np.arange(BLOCK_SIZE * 2, dtype=np.float32)
    in_r = np.zeros(BLOCK_SIZE * 2, np.float32)
    out_l, _ = e.process_block(in_l, in_r)
    expected = np.zeros(BLOCK_SIZE * 2, np.float32)
    expected[delay:] = in_l[: BLOCK_SIZE * 2 - delay]
    _assert_approx(out_l[delay:], expected[delay:], "Delayed signal")


def test_delay_ir_at_the_tolerance_f32_allows(oracle):
    """the same ramp at 5e-4 absolute / 1e-6 relative RMS (the reference's 1e-3 cannot tell a correct FFT from a sloppy
    one): the checker itself must meet what the HIP path is held to"""
    e = oracle.ConvolutionEngine()
    delay = 5
    ir = np.zeros(delay + 1, np.float32)
    ir[delay] = 1.0
    e.set_ir(oracle.LSL, ir)
    for p in (oracle.LSR, oracle.RSL, oracle.RSR):
        e.set_ir(p, [0.0])
    in_l = np.arange(BLOCK_SIZE * 2, dtype=np.float32)
    out_l, out_r = e.process_block(in_l, np.zeros(BLOCK_SIZE * 2, np.float32))
    expected = np.zeros(BLOCK_SIZE * 2)
    expected[delay:] = in_l[: BLOCK_SIZE * 2 - delay]
    err = np.abs(out_l - expected)
    assert err.max() <= 5e-4 and np.abs(out_r).max() <= 5e-4
    assert np.sqrt(np.mean(err ** 2)) / np.sqrt(np.mean(expected ** 2)) <= 1e-6


def test_long_ir_partitioning(oracle):
    e = oracle.ConvolutionEngine()
    ir_len = BLOCK_SIZE + BLOCK_SIZE // 2
    ir = np.zeros(ir_len, np.float32)
    ir[0] = 1.0
    ir[ir_len - 1] = 0.5
    e.set_ir(oracle.LSL, ir)
    assert e.num_partitions(oracle.LSL) == 2, "IR should be split into 2 partitions"
    in_l = np.zeros(BLOCK_SIZE * 3, np.float32)
    in_l[0] = 1.0
    in_r = np.zeros(BLOCK_SIZE * 3, np.float32)
    out_l, _ = e.process_block(in_l, in_r)
    expected = np.zeros(BLOCK_SIZE * 3, np.float32)
    expected[0] = 1.0
    expected[ir_len - 1] = 0.5
    _assert_approx(out_l[:ir_len], expected[:ir_len], "Long IR convolution")


def test_biquad_filter_passthrough_when_disabled(oracle):
    # BiquadFilter::new(48000) is disabled: process_sample(0.5) == 0.5 bit-exactly
    eq = oracle.StereoParametricEQ(1, 48000.0)
    l = np.array([0.5], np.float32)
    r = np.array([0.5], np.float32)
    eq.process_block(l, r)
    assert l[0] == np.float32(0.5) and r[0] == np.float32(0.5)


def test_biquad_filter_processes_when_enabled(oracle):
    eq = oracle.StereoParametricEQ(1, 48000.0)
    eq.update_band_coeffs(0, 48000.0, oracle.LOWPASS, 1000.0, 0.707, 0.0, True)
    l = np.array([0.5], np.float32)
    r = np.array([0.5], np.float32)
    eq.process_block(l, r)
    assert l[0] != np.float32(0.5)
