"""open_headstage_amd -- MI355X-native binaural convolution core.

Drop-in for the `src/dsp` hot path of KiloHertzian/Open-Headstage (4-path HRIR
partitioned FFT convolution + 10-band parametric EQ), as hand-written gfx950 HIP
kernels behind the C ABI in include/ohs_hip.h.  This package is the host-side
mirror of the reference's Rust interface plus the many-stream batch mode; it
contains no CPU compute path.
"""
from .dsp import (BLOCK_SIZE, FFT_SIZE, NUM_EQ_BANDS, BandConfig, BiquadFilter, ConvolutionEngine, ConvolutionPath,
                  FilterType, StereoParametricEQ, biquad_coefficients, process_chain)
from .batch import BatchProcessor, NodeBatchProcessor
from .autoeq import BandSetting, apply_bands, parse_autoeq_csv, parse_autoeq_csv_text
from ._ffi import OhsError

__all__ = ["BLOCK_SIZE", "FFT_SIZE", "NUM_EQ_BANDS", "BandConfig", "BiquadFilter", "ConvolutionEngine",
           "ConvolutionPath", "FilterType", "StereoParametricEQ", "biquad_coefficients",
           "process_chain", "BatchProcessor", "NodeBatchProcessor", "OhsError", "BandSetting", "apply_bands",
           "parse_autoeq_csv", "parse_autoeq_csv_text"]
