"""Host-side mirror of the reference's `src/dsp` interface over the C ABI.

Same names, argument meaning and error behaviour as the Rust types so that the
parity tests read like the reference's own tests:

  ConvolutionPath        src/dsp/convolution.rs:28-33
  ConvolutionEngine      src/dsp/convolution.rs:68-295    new / set_ir / process_block / clone
  FilterType, BandConfig src/dsp/parametric_eq.rs:23-44
  StereoParametricEQ     src/dsp/parametric_eq.rs:125-210
  process_chain          src/lib.rs:1169-1207             (EQ -> convolution -> gain)

Where the reference panics (slice-length mismatch convolution.rs:164; `unwrap`
on OutsideNyquist / NegativeQ parametric_eq.rs:111) these raise.
All arithmetic runs in the HIP kernels behind libohs_hip.so.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass

import numpy as np

from . import _ffi
from ._ffi import check, fp, lib

BLOCK_SIZE = 512   # convolution.rs:22
FFT_SIZE = 1024    # convolution.rs:23
NUM_EQ_BANDS = 10  # lib.rs:40


class ConvolutionPath(enum.IntEnum):
    Lsl = 0
    Lsr = 1
    Rsl = 2
    Rsr = 3


class FilterType(enum.IntEnum):
    Peak = 0
    LowShelf = 1
    HighShelf = 2
    LowPass = 3
    HighPass = 4
    BandPass = 5
    Notch = 6
    AllPass = 7


@dataclass
class BandConfig:
    filter_type: FilterType = FilterType.Peak
    center_freq: float = 1000.0
    q: float = 0.7
    gain_db: float = 0.0
    enabled: bool = False


def _in(a) -> np.ndarray:
    if type(a) is np.ndarray and a.dtype == np.float32 and a.ndim == 1 and a.flags.c_contiguous:
        return a            # the per-call path of an audio host: nothing to convert
    return np.ascontiguousarray(a, dtype=np.float32).ravel()


def _addr(a: np.ndarray) -> int:
    """Address of a contiguous array for the per-call entry points (their prototypes take void*): five times cheaper
    than building a ctypes pointer object per argument."""
    return a.__array_interface__["data"][0]


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(fp)


def _out(a, n: int, name: str) -> np.ndarray:
    if a is None:
        return np.empty(n, np.float32)
    if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.c_contiguous and a.ndim == 1):
        raise TypeError(f"{name} must be a contiguous 1-D float32 ndarray")
    if a.size != n:   # copy_from_slice panics on a length mismatch (convolution.rs:164,170)
        raise ValueError(f"{name} has {a.size} frames, input has {n}")
    return a


class ConvolutionEngine:
    def __init__(self, device: int = 0, _handle=None):
        self.device = device
        if _handle is None:
            h = C.c_void_p()
            check(lib().ohs_engine_create(device, C.byref(h)))
            _handle = h
        self._h = _handle
        self._process = lib().ohs_engine_process_block

    @classmethod
    def new(cls, device: int = 0) -> "ConvolutionEngine":
        return cls(device)

    def clone(self) -> "ConvolutionEngine":
        h = C.c_void_p()
        check(lib().ohs_engine_clone(self._h, C.byref(h)))
        return ConvolutionEngine(self.device, h)

    def set_ir(self, path, ir_data) -> None:
        ir = _in(ir_data)
        check(lib().ohs_engine_set_ir(self._h, int(path), _ptr(ir), ir.size))

    def set_speakers(self, sofa, az_l: float = -30.0, el_l: float = 0.0, az_r: float = 30.0, el_r: float = 0.0,
                     radius_m: float = 1.0, fs: float = 0.0) -> int:
        """The plugin's speaker angles (CLAP az_l / el_l / az_r / el_r, degrees, azimuth positive to the right;
        src/lib.rs:120-128) -> two get_hrtf_irs -> set_ir of the paths whose response changed
        (ohs_engine_set_speakers).  Returns the bit mask of the paths that were re-loaded."""
        m = C.c_uint()
        check(lib().ohs_engine_set_speakers(self._h, sofa._h, az_l, el_l, az_r, el_r, radius_m, fs, C.byref(m)))
        return int(m.value)

    def num_partitions(self, path) -> int:
        n = C.c_size_t()
        check(lib().ohs_engine_num_partitions(self._h, int(path), C.byref(n)))
        return int(n.value)

    def set_realtime(self, enable: bool) -> None:
        """Resident-kernel mode for real-time sized calls (ohs_engine_set_realtime)."""
        check(lib().ohs_engine_set_realtime(self._h, int(bool(enable))))

    def set_flush_denormals(self, mode: int) -> None:
        """0 = IEEE denormals (default), 1 = flush denormal results (nih-plug's FTZ scope around Plugin::process,
        src/lib.rs:1156), 2 = also read denormal inputs as zero (FTZ | DAZ).  ohs_engine_set_flush_denormals."""
        check(lib().ohs_engine_set_flush_denormals(self._h, int(mode)))

    def latency_samples(self, n: int) -> int:
        """Delay of the FIFO adapter for a constant host block size n (convolution.rs:163-181)."""
        v = C.c_size_t()
        check(lib().ohs_engine_latency_samples(self._h, int(n), C.byref(v)))
        return int(v.value)

    def silent_frames(self) -> int:
        v = C.c_uint64()
        check(lib().ohs_engine_silent_frames(self._h, C.byref(v)))
        return int(v.value)

    def process_block(self, input_left, input_right, output_left=None, output_right=None):
        il, ir = _in(input_left), _in(input_right)
        if il.size != ir.size:
            raise ValueError("input_left / input_right length mismatch")
        ol = _out(output_left, il.size, "output_left")
        orr = _out(output_right, il.size, "output_right")
        check(self._process(self._h, _addr(il), _addr(ir), _addr(ol), _addr(orr), il.size))
        return ol, orr

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().ohs_engine_destroy(h)
            except Exception:
                pass


def biquad_coefficients(filter_type, sample_rate: float, center_freq: float, q: float,
                        gain_db: float) -> np.ndarray:
    """biquad::Coefficients::<f32>::from_params -> [b0, b1, b2, a1, a2]."""
    out = np.zeros(5, np.float32)
    check(lib().ohs_biquad_coeffs(int(filter_type), sample_rate, center_freq, q, gain_db, _ptr(out)))
    return out


class BiquadFilter:
    """`BiquadFilter` (parametric_eq.rs:46-123) over ohs_biquad_*: one DF2T section, disabled until set_enabled(True)."""

    def __init__(self, initial_sample_rate: float, device: int = 0, _handle=None):
        self.device = device
        if _handle is None:
            h = C.c_void_p()
            check(lib().ohs_biquad_create(device, initial_sample_rate, C.byref(h)))
            _handle = h
        self._h = _handle

    @classmethod
    def new(cls, initial_sample_rate: float, device: int = 0) -> "BiquadFilter":
        return cls(initial_sample_rate, device)

    def clone(self) -> "BiquadFilter":
        h = C.c_void_p()
        check(lib().ohs_biquad_clone(self._h, C.byref(h)))
        return BiquadFilter(0.0, self.device, h)

    def reset_state(self) -> None:
        check(lib().ohs_biquad_reset_state(self._h))

    def set_enabled(self, enabled: bool) -> None:
        check(lib().ohs_biquad_set_enabled(self._h, int(bool(enabled))))

    def update_coeffs(self, filter_type, sample_rate: float, center_freq: float, q: float, gain_db: float) -> None:
        check(lib().ohs_biquad_update_coeffs(self._h, int(filter_type), sample_rate, center_freq, q, gain_db))

    def set_coeffs(self, coeffs) -> None:
        c = _in(coeffs)
        if c.size != 5:
            raise ValueError("coeffs must be [b0, b1, b2, a1, a2]")
        check(lib().ohs_biquad_set_coeffs(self._h, _ptr(c)))

    def process_sample(self, input_sample: float) -> float:
        a = np.array([input_sample], np.float32)
        check(lib().ohs_biquad_process(self._h, _addr(a), 1))
        return float(a[0])

    def process(self, samples: np.ndarray) -> None:
        """n consecutive process_sample calls, in place"""
        if not (isinstance(samples, np.ndarray) and samples.dtype == np.float32 and samples.flags.c_contiguous and samples.ndim == 1):
            raise TypeError("samples must be a contiguous 1-D float32 ndarray")
        check(lib().ohs_biquad_process(self._h, _addr(samples), samples.size))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().ohs_biquad_destroy(h)
            except Exception:
                pass


class StereoParametricEQ:
    def __init__(self, num_bands: int, initial_sample_rate: float, device: int = 0):
        self.num_bands = int(num_bands)
        self.device = device
        h = C.c_void_p()
        check(lib().ohs_eq_create(device, self.num_bands, initial_sample_rate, C.byref(h)))
        self._h = h

    @classmethod
    def new(cls, num_bands: int, initial_sample_rate: float, device: int = 0):
        return cls(num_bands, initial_sample_rate, device)

    def update_band_coeffs(self, band_idx: int, sample_rate: float, config: BandConfig) -> None:
        check(lib().ohs_eq_update_band(self._h, int(band_idx), sample_rate, int(config.filter_type),
                                       config.center_freq, config.q, config.gain_db,
                                       int(bool(config.enabled))))

    def set_band_coeffs(self, band_idx: int, coeffs, enabled: bool) -> None:
        c = _in(coeffs)
        if c.size != 5:
            raise ValueError("coeffs must be [b0, b1, b2, a1, a2]")
        check(lib().ohs_eq_set_band_coeffs(self._h, int(band_idx), _ptr(c), int(bool(enabled))))

    def get_band_coeffs(self, band_idx: int):
        c = np.zeros(5, np.float32)
        en = C.c_int()
        check(lib().ohs_eq_get_band_coeffs(self._h, int(band_idx), _ptr(c), C.byref(en)))
        return c, bool(en.value)

    def process_block(self, input_left: np.ndarray, input_right: np.ndarray) -> None:
        """In place, like `process_block(&mut [f32], &mut [f32])`."""
        for a in (input_left, input_right):
            if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.c_contiguous):
                raise TypeError("in-place EQ needs contiguous float32 ndarrays")
        if input_left.size != input_right.size:
            raise ValueError("left / right length mismatch")
        check(lib().ohs_eq_process_block(self._h, _addr(input_left), _addr(input_right), input_left.size))

    def set_exact_specials(self, enable: bool) -> None:
        """Opt into bit-exact handling of -0.0 and of samples around a non-finite input (ohs_eq_set_exact_specials)."""
        check(lib().ohs_eq_set_exact_specials(self._h, int(bool(enable))))

    def set_flush_denormals(self, mode: int) -> None:
        """see ConvolutionEngine.set_flush_denormals (ohs_eq_set_flush_denormals)"""
        check(lib().ohs_eq_set_flush_denormals(self._h, int(mode)))

    def reset_all_bands_state(self) -> None:
        check(lib().ohs_eq_reset(self._h))

    def calculate_frequency_response(self, sample_rate: float, frequencies) -> np.ndarray:
        f = _in(frequencies)
        out = np.empty(f.size, np.float32)
        check(lib().ohs_eq_frequency_response(self._h, sample_rate, _ptr(f), f.size, _ptr(out)))
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().ohs_eq_destroy(h)
            except Exception:
                pass


def process_chain(engine: ConvolutionEngine, eq: StereoParametricEQ | None, left: np.ndarray,
                  right: np.ndarray, *, master_bypass: bool = False, eq_enable: bool = False,
                  output_gain: float = 1.0) -> None:
    """The DSP part of `Plugin::process` (lib.rs:1169-1207), in place on left/right."""
    for a in (left, right):
        if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.c_contiguous):
            raise TypeError("process_chain needs contiguous float32 ndarrays")
    if left.size != right.size:
        raise ValueError("Mismatched channel length")
    check(lib().ohs_chain_process(engine._h, eq._h if eq is not None else None, int(master_bypass),
                                  int(eq_enable), output_gain, _addr(left), _addr(right), left.size))
