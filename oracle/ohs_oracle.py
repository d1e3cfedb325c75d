"""ctypes binding of the CPU ORACLE (oracle/ohs_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from open_headstage_amd/ (the product).
The classes mirror the reference's names (src/dsp/convolution.rs,
src/dsp/parametric_eq.rs) so the known-answer tests read like the reference's.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libohs_oracle.so")

BLOCK_SIZE = 512
FFT_SIZE = 1024

LSL, LSR, RSL, RSR = 0, 1, 2, 3
PEAK, LOWSHELF, HIGHSHELF, LOWPASS, HIGHPASS, BANDPASS, NOTCH, ALLPASS = range(8)


def build(force: bool = False) -> str:
    """Compile oracle/ohs_oracle.c -> libohs_oracle.so (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "ohs_oracle.c")
    hdr = os.path.join(_HERE, "ohs_oracle.h")
    libm = os.path.join(_HERE, "libm_musl.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr, libm)
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libohs_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


def build_native() -> str:
    """The same source with -march=native, compiled ON THE MACHINE THAT RUNS IT (bench.py's cpu_baseline leg
    on the GPU box: the north_star asks for the CPU path timed on that box's host cores).  The portable
    -mavx2 library stays what the parity tests load; rounding is identical (-ffp-contract=off, no fast-math),
    only the vectorisation differs."""
    out = os.path.join(_HERE, "libohs_oracle_native.so")
    src = os.path.join(_HERE, "ohs_oracle.c")
    libm = os.path.join(_HERE, "libm_musl.h")
    if not os.path.exists(out) or max(os.path.getmtime(src), os.path.getmtime(libm)) > os.path.getmtime(out):
        subprocess.run(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c11",
                        "-shared", "-o", out, src, "-lm"], check=True)
    return out


def use_library(path: str | None) -> None:
    """Select the shared object lib() loads (None = the portable one); call before creating oracle objects."""
    global _lib, _SO_OVERRIDE
    _SO_OVERRIDE = path
    _lib = None


_lib = None
_SO_OVERRIDE = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if _SO_OVERRIDE is None and not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO_OVERRIDE or _SO)
    fp = C.POINTER(C.c_float)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int)
    L.ohs_or_fft1024.argtypes = [fp, C.c_int]
    L.ohs_or_engine_new.restype = C.c_void_p
    L.ohs_or_engine_clone.restype = C.c_void_p
    L.ohs_or_engine_clone.argtypes = [C.c_void_p]
    L.ohs_or_engine_free.argtypes = [C.c_void_p]
    L.ohs_or_engine_set_ir.argtypes = [C.c_void_p, C.c_int, fp, C.c_size_t]
    L.ohs_or_engine_process_block.argtypes = [C.c_void_p, fp, fp, fp, fp, C.c_size_t]
    L.ohs_or_engine_num_partitions.restype = C.c_size_t
    L.ohs_or_engine_num_partitions.argtypes = [C.c_void_p, C.c_int]
    L.ohs_or_biquad_coeffs.restype = C.c_int
    L.ohs_or_biquad_coeffs.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, fp]
    L.ohs_or_eq_new.restype = C.c_void_p
    L.ohs_or_eq_new.argtypes = [C.c_size_t, C.c_float]
    L.ohs_or_eq_free.argtypes = [C.c_void_p]
    L.ohs_or_eq_update_band.restype = C.c_int
    L.ohs_or_eq_update_band.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_float,
                                        C.c_float, C.c_float, C.c_int]
    L.ohs_or_eq_set_band_coeffs.argtypes = [C.c_void_p, C.c_size_t, fp, C.c_int]
    L.ohs_or_eq_get_band_coeffs.argtypes = [C.c_void_p, C.c_size_t, fp, ip]
    L.ohs_or_eq_process_block.argtypes = [C.c_void_p, fp, fp, C.c_size_t]
    L.ohs_or_eq_reset.argtypes = [C.c_void_p]
    L.ohs_or_eq_frequency_response.argtypes = [C.c_void_p, C.c_float, fp, C.c_size_t, fp]
    L.ohs_or_chain_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, fp, fp,
                                       C.c_size_t]
    L.ohs_or_set_flush_denormals.restype = C.c_int
    L.ohs_or_set_flush_denormals.argtypes = [C.c_int]
    L.ohs_or_direct_conv_f64.argtypes = [fp, C.c_size_t, fp, C.c_size_t, dp]
    L.ohs_or_binaural_f64.argtypes = [dp, dp, C.c_size_t, fp, C.c_size_t, fp, C.c_size_t, fp,
                                      C.c_size_t, fp, C.c_size_t, dp, dp]
    L.ohs_or_eq_f64.argtypes = [fp, ip, C.c_size_t, fp, C.c_size_t, dp, dp]
    _lib = L
    return L


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def fft1024(z: np.ndarray, inverse: bool = False) -> np.ndarray:
    buf = np.ascontiguousarray(z, dtype=np.complex64).copy()
    assert buf.shape == (FFT_SIZE,)
    lib().ohs_or_fft1024(buf.view(np.float32).ctypes.data_as(C.POINTER(C.c_float)), int(inverse))
    return buf


class ConvolutionEngine:
    """Oracle twin of convolution.rs `ConvolutionEngine` (:68-295)."""

    def __init__(self, _handle=None):
        self._h = _handle if _handle is not None else lib().ohs_or_engine_new()

    def clone(self) -> "ConvolutionEngine":
        return ConvolutionEngine(lib().ohs_or_engine_clone(self._h))

    def set_ir(self, path: int, ir) -> None:
        ir = _f32(ir).ravel()
        lib().ohs_or_engine_set_ir(self._h, int(path), _fp(ir), ir.size)

    def num_partitions(self, path: int) -> int:
        return int(lib().ohs_or_engine_num_partitions(self._h, int(path)))

    def process_block(self, in_l, in_r, out_l=None, out_r=None):
        in_l = _f32(in_l); in_r = _f32(in_r)
        n = in_l.size
        assert in_r.size == n
        if out_l is None:
            out_l = np.empty(n, np.float32)
        if out_r is None:
            out_r = np.empty(n, np.float32)
        assert out_l.size == n and out_r.size == n and out_l.dtype == np.float32
        lib().ohs_or_engine_process_block(self._h, _fp(in_l), _fp(in_r), _fp(out_l), _fp(out_r), n)
        return out_l, out_r

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ohs_or_engine_free(self._h)
            self._h = None


class flush_denormals:
    """Context manager: run the oracle calls inside with the calling thread's MXCSR set as the shipped plugin runs
    (mode 1 = FTZ, nih-plug's scope around Plugin::process, src/lib.rs:1156; 2 = FTZ | DAZ), restored on exit.
    numpy arithmetic on this thread is affected too while the mode is set: keep the block to oracle calls."""

    def __init__(self, mode: int):
        self.mode = int(mode)

    def __enter__(self):
        self.prev = lib().ohs_or_set_flush_denormals(self.mode)
        if self.prev < 0:
            raise RuntimeError("no flush-to-zero control on this CPU")
        return self

    def __exit__(self, *exc):
        lib().ohs_or_set_flush_denormals(self.prev)
        return False


def biquad_coeffs(filter_type: int, fs: float, fc: float, q: float, gain_db: float):
    out = np.zeros(5, np.float32)
    rc = lib().ohs_or_biquad_coeffs(int(filter_type), fs, fc, q, gain_db, _fp(out))
    if rc:
        raise ValueError({1: "OutsideNyquist", 2: "NegativeQ"}[rc])
    return out


class StereoParametricEQ:
    """Oracle twin of parametric_eq.rs `StereoParametricEQ` (:125-210)."""

    def __init__(self, num_bands: int, fs: float):
        self.num_bands = int(num_bands)
        self._h = lib().ohs_or_eq_new(self.num_bands, fs)

    def update_band_coeffs(self, band, fs, filter_type, fc, q, gain_db, enabled):
        rc = lib().ohs_or_eq_update_band(self._h, int(band), fs, int(filter_type), fc, q, gain_db,
                                         int(bool(enabled)))
        if rc:
            raise ValueError({1: "OutsideNyquist", 2: "NegativeQ"}[rc])

    def set_band_coeffs(self, band, coeffs, enabled):
        c = _f32(coeffs)
        lib().ohs_or_eq_set_band_coeffs(self._h, int(band), _fp(c), int(bool(enabled)))

    def get_band_coeffs(self, band):
        c = np.zeros(5, np.float32)
        en = C.c_int(0)
        lib().ohs_or_eq_get_band_coeffs(self._h, int(band), _fp(c), C.byref(en))
        return c, bool(en.value)

    def process_block(self, l: np.ndarray, r: np.ndarray):
        assert l.dtype == np.float32 and r.dtype == np.float32 and l.size == r.size
        assert l.flags.c_contiguous and r.flags.c_contiguous
        lib().ohs_or_eq_process_block(self._h, _fp(l), _fp(r), l.size)

    def reset_all_bands_state(self):
        lib().ohs_or_eq_reset(self._h)

    def calculate_frequency_response(self, fs, freqs):
        f = _f32(freqs)
        out = np.empty(f.size, np.float32)
        lib().ohs_or_eq_frequency_response(self._h, fs, _fp(f), f.size, _fp(out))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ohs_or_eq_free(self._h)
            self._h = None


def chain_process(engine: ConvolutionEngine, eq: StereoParametricEQ, l: np.ndarray, r: np.ndarray,
                  *, bypass=False, eq_enable=True, gain=1.0):
    """lib.rs:1169-1207 in place on l, r."""
    assert l.dtype == np.float32 and r.dtype == np.float32 and l.size == r.size
    lib().ohs_or_chain_process(engine._h, eq._h, int(bypass), int(eq_enable), gain, _fp(l), _fp(r),
                               l.size)


# ---- f64 ground truth --------------------------------------------------------

def direct_conv_f64(x, h) -> np.ndarray:
    x = _f32(x); h = _f32(h)
    y = np.empty(x.size, np.float64)
    lib().ohs_or_direct_conv_f64(_fp(x), x.size, _fp(h), h.size, _dp(y))
    return y


def binaural_f64(x_l, x_r, irs):
    """irs = [lsl, lsr, rsl, rsr]; inputs may be float64 (e.g. f64 EQ output)."""
    x_l = np.ascontiguousarray(x_l, np.float64); x_r = np.ascontiguousarray(x_r, np.float64)
    hs = [_f32(h).ravel() for h in irs]
    n = x_l.size
    out_l = np.empty(n, np.float64); out_r = np.empty(n, np.float64)
    lib().ohs_or_binaural_f64(_dp(x_l), _dp(x_r), n, _fp(hs[0]), hs[0].size, _fp(hs[1]), hs[1].size,
                              _fp(hs[2]), hs[2].size, _fp(hs[3]), hs[3].size, _dp(out_l), _dp(out_r))
    return out_l, out_r


def eq_f64(coeffs, enabled, x, state=None):
    coeffs = _f32(coeffs).reshape(-1, 5)
    nb = coeffs.shape[0]
    en = np.ascontiguousarray(enabled, np.int32)
    x = _f32(x)
    if state is None:
        state = np.zeros(2 * nb, np.float64)
    y = np.empty(x.size, np.float64)
    lib().ohs_or_eq_f64(_fp(coeffs), en.ctypes.data_as(C.POINTER(C.c_int)), nb, _fp(x), x.size,
                        _dp(state), _dp(y))
    return y, state
