"""The reference's own unit tests for the hot path, run against the HIP path through the C ABI.

Same inputs, same expectations and the same 1e-3 tolerance as
  src/dsp/convolution.rs:317-347, 349-383, 385-421   and
  src/dsp/parametric_eq.rs:218-225, 227-238
(the oracle twin of this file is tests/test_oracle_reference_kats.py).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BLOCK_SIZE = 512
TOLERANCE = 1e-3


def _assert_approx(a, b, msg):
    assert a.shape == b.shape, msg
    assert np.all(np.abs(a - b) < TOLERANCE), f"{msg}: max err {np.abs(a - b).max()}"


def test_identity_ir_passthrough():
    from open_headstage_amd import ConvolutionEngine, ConvolutionPath
    engine = ConvolutionEngine.new()
    engine.set_ir(ConvolutionPath.Lsl, [1.0])
    engine.set_ir(ConvolutionPath.Lsr, [0.0])
    engine.set_ir(ConvolutionPath.Rsl, [0.0])
    engine.set_ir(ConvolutionPath.Rsr, [1.0])
    i = np.arange(BLOCK_SIZE, dtype=np.float32)
    input_l = np.sin(i * np.float32(0.1)).astype(np.float32)
    input_r = np.sin(i * np.float32(-0.1)).astype(np.float32)
    output_l = np.zeros(BLOCK_SIZE, np.float32)
    output_r = np.zeros(BLOCK_SIZE, np.float32)
    engine.process_block(input_l, input_r, output_l, output_r)
    engine.process_block(input_l, input_r, output_l, output_r)
    _assert_approx(output_l, input_l, "Identity passthrough L channel")
    _assert_approx(output_r, input_r, "Identity passthrough R channel")


def test_delay_ir():
    from open_headstage_amd import ConvolutionEngine, ConvolutionPath
    engine = ConvolutionEngine.new()
    delay_samples = 5
    ir = np.zeros(delay_samples + 1, np.float32)
    ir[delay_samples] = 1.0
    engine.set_ir(ConvolutionPath.Lsl, ir)
    engine.set_ir(ConvolutionPath.Lsr, [0.0])
    engine.set_ir(ConvolutionPath.Rsl, [0.0])
    engine.set_ir(ConvolutionPath.Rsr, [0.0])
    input_l = np.arange(BLOCK_SIZE * 2, dtype=np.float32)
    input_r = np.zeros(BLOCK_SIZE * 2, np.float32)
    output_l, output_r = engine.process_block(input_l, input_r)
    expected = np.zeros(BLOCK_SIZE * 2, np.float32)
    expected[delay_samples:] = input_l[: BLOCK_SIZE * 2 - delay_samples]
    _assert_approx(output_l[delay_samples:], expected[delay_samples:], "Delayed signal")
    # the muted right ear only carries rounding leakage of the packed L/R transform
    assert np.all(np.abs(output_r) < TOLERANCE)


def test_long_ir_partitioning():
    from open_headstage_amd import ConvolutionEngine, ConvolutionPath
    engine = ConvolutionEngine.new()
    ir_len = BLOCK_SIZE + BLOCK_SIZE // 2
    ir = np.zeros(ir_len, np.float32)
    ir[0] = 1.0
    ir[ir_len - 1] = 0.5
    engine.set_ir(ConvolutionPath.Lsl, ir)
    assert engine.num_partitions(ConvolutionPath.Lsl) == 2, "IR should be split into 2 partitions"
    input_l = np.zeros(BLOCK_SIZE * 3, np.float32)
    input_l[0] = 1.0
    input_r = np.zeros(BLOCK_SIZE * 3, np.float32)
    output_l, _ = engine.process_block(input_l, input_r)
    expected = np.zeros(BLOCK_SIZE * 3, np.float32)
    expected[0] = 1.0
    expected[ir_len - 1] = 0.5
    _assert_approx(output_l[:ir_len], expected[:ir_len], "Long IR convolution")


def test_biquad_filter_passthrough_when_disabled():
    from open_headstage_amd import StereoParametricEQ
    eq = StereoParametricEQ.new(1, 48000.0)     # a fresh band is disabled (parametric_eq.rs:74)
    l = np.array([0.5], np.float32)
    r = np.array([0.5], np.float32)
    eq.process_block(l, r)
    assert l[0] == np.float32(0.5) and r[0] == np.float32(0.5), "Filter should be passthrough when disabled"


def test_biquad_filter_processes_when_enabled():
    from open_headstage_amd import BandConfig, FilterType, StereoParametricEQ
    eq = StereoParametricEQ.new(1, 48000.0)
    eq.update_band_coeffs(0, 48000.0, BandConfig(FilterType.LowPass, 1000.0, 0.707, 0.0, True))
    l = np.array([0.5], np.float32)
    r = np.array([0.5], np.float32)
    eq.process_block(l, r)
    assert l[0] != np.float32(0.5), "Filter should process the sample when enabled"
