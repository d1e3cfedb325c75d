"""SOFA HRIR source -- host-side mirror of src/sofa/loader.rs `MySofa` and the
speaker -> path wiring the reference plans but does not have
(github_issues/sofa_implement_logic_select_extract_hrirs.md:5): the left speaker's (left-ear,
right-ear) responses become Lsl / Lsr, the right speaker's Rsl / Rsr."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from ._ffi import check, fp, lib
from .dsp import ConvolutionPath


# ohs_sofa_open_ex flags (include/ohs_hip.h)
SOFA_RESAMPLE, SOFA_LOUDNESS, SOFA_INTERPOLATE, SOFA_MYSOFA_DEFAULTS = 1, 2, 4, 7


class MySofa:
    def __init__(self, filepath, flags: int = 0, target_samplerate: float = 0.0):
        h = C.c_void_p()
        if flags:
            check(lib().ohs_sofa_open_ex(os.fsencode(filepath), int(flags), float(target_samplerate), C.byref(h)))
        else:
            check(lib().ohs_sofa_open(os.fsencode(filepath), C.byref(h)))
        self._h = h
        self.flags = int(flags)
        M, R, N, fs = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_float()
        check(lib().ohs_sofa_info(h, C.byref(M), C.byref(R), C.byref(N), C.byref(fs)))
        self.num_measurements, self.num_receivers = int(M.value), int(R.value)
        self.filter_length = int(N.value)                 # loader.rs:102
        self.source_samplerate = float(fs.value)          # loader.rs:103-118

    @classmethod
    def open(cls, filepath, target_samplerate: float | None = None) -> "MySofa":
        s = cls(filepath)
        s.target_samplerate = target_samplerate
        return s

    @classmethod
    def open_like_mysofa(cls, filepath, target_samplerate: float, flags: int = SOFA_MYSOFA_DEFAULTS) -> "MySofa":
        """What the reference's MySofa::open gets from mysofa_open (loader.rs:83-90): data resampled to the
        target rate, loudness-normalised, get_hrtf_irs interpolating between neighbours.  Parity unpinned
        (csrc/sofa_conditioning.cpp)."""
        s = cls(filepath, flags, target_samplerate)
        s.target_samplerate = None          # already at the target rate
        return s

    def loudness_factor(self) -> float:
        v = C.c_float()
        check(lib().ohs_sofa_loudness_factor(self._h, C.byref(v)))
        return float(v.value)

    def neighbors(self, m: int):
        out = (C.c_int * 6)()
        check(lib().ohs_sofa_neighbors(self._h, int(m), out))
        return [int(v) for v in out]

    @staticmethod
    def spherical_to_cartesian(spherical) -> np.ndarray:      # loader.rs:203-213
        a = np.ascontiguousarray(spherical, np.float32); o = np.zeros(3, np.float32)
        check(lib().ohs_sofa_spherical_to_cartesian(a.ctypes.data_as(fp), o.ctypes.data_as(fp)))
        return o

    @staticmethod
    def cartesian_to_spherical(cartesian) -> np.ndarray:      # loader.rs:218-229
        a = np.ascontiguousarray(cartesian, np.float32); o = np.zeros(3, np.float32)
        check(lib().ohs_sofa_cartesian_to_spherical(a.ctypes.data_as(fp), o.ctypes.data_as(fp)))
        return o

    def position(self, m: int) -> np.ndarray:
        out = np.zeros(3, np.float32)
        check(lib().ohs_sofa_position(self._h, int(m), out.ctypes.data_as(fp)))
        return out

    def nearest(self, azimuth_deg: float, elevation_deg: float, radius_m: float = 1.0) -> int:
        i = C.c_size_t()
        check(lib().ohs_sofa_nearest(self._h, azimuth_deg, elevation_deg, radius_m, C.byref(i)))
        return int(i.value)

    def ir(self, m: int, receiver: int) -> np.ndarray:
        out = np.zeros(self.filter_length, np.float32)
        check(lib().ohs_sofa_get_ir(self._h, int(m), int(receiver), out.ctypes.data_as(fp), out.size))
        return out

    def get_hrtf_irs(self, azimuth_deg: float, elevation_deg: float, radius_m: float = 1.0,
                     target_samplerate: float | None = None):
        """-> (left_ir, right_ir) of the nearest measurement (loader.rs:136-199), optionally resampled."""
        l = np.zeros(self.filter_length, np.float32)
        r = np.zeros(self.filter_length, np.float32)
        dl, dr = C.c_float(), C.c_float()
        check(lib().ohs_sofa_get_hrtf_irs(self._h, azimuth_deg, elevation_deg, radius_m, l.ctypes.data_as(fp),
                                          r.ctypes.data_as(fp), l.size, C.byref(dl), C.byref(dr)))
        fs = target_samplerate or getattr(self, "target_samplerate", None)
        if fs and abs(fs - self.source_samplerate) > 1e-3:
            l, r = resample_ir(l, self.source_samplerate, fs), resample_ir(r, self.source_samplerate, fs)
        return l, r

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().ohs_sofa_close(h)

    __del__ = close


def resample_ir(ir, fs_in: float, fs_out: float) -> np.ndarray:
    x = np.ascontiguousarray(ir, np.float32)
    n = C.c_size_t()
    check(lib().ohs_sofa_resample_ir(x.ctypes.data_as(fp), x.size, fs_in, fs_out, None, 0, C.byref(n)))
    out = np.zeros(n.value, np.float32)
    check(lib().ohs_sofa_resample_ir(x.ctypes.data_as(fp), x.size, fs_in, fs_out, out.ctypes.data_as(fp),
                                     out.size, C.byref(n)))
    return out


def speaker_hrirs(sofa: MySofa, az_left: float = 30.0, el_left: float = 0.0, az_right: float = -30.0,
                  el_right: float = 0.0, radius_m: float = 1.0, target_samplerate: float | None = None):
    """[lsl, lsr, rsl, rsr] for two virtual speakers.  AES69 azimuth is positive to the LEFT, so the
    plugin's az_l = -30 / az_r = +30 (lib.rs:120-128, clockwise-positive UI angles) map to +30 / -30 here."""
    lsl, lsr = sofa.get_hrtf_irs(az_left, el_left, radius_m, target_samplerate)
    rsl, rsr = sofa.get_hrtf_irs(az_right, el_right, radius_m, target_samplerate)
    return [lsl, lsr, rsl, rsr]


def speaker_irs_plugin_angles(sofa: MySofa, az_l: float = -30.0, el_l: float = 0.0, az_r: float = 30.0, el_r: float = 0.0,
                              radius_m: float = 1.0, fs: float = 0.0):
    """[lsl, lsr, rsl, rsr] as ohs_*_set_speakers computes them inside the library (ohs_sofa_speaker_irs): the PLUGIN's
    angles (degrees, azimuth positive to the right -- lib.rs:120-128), resampled to fs when fs > 0."""
    lens = (C.c_size_t * 4)()
    check(lib().ohs_sofa_speaker_irs(sofa._h, az_l, el_l, az_r, el_r, radius_m, fs, None, 0, lens))
    outs = [np.zeros(int(n), np.float32) for n in lens]
    ptrs = (fp * 4)(*[o.ctypes.data_as(fp) for o in outs])
    check(lib().ohs_sofa_speaker_irs(sofa._h, az_l, el_l, az_r, el_r, radius_m, fs, ptrs, max(int(n) for n in lens), lens))
    return outs


def load_into(target, irs) -> None:
    """get_hrtf_irs -> set_ir wiring: target is a ConvolutionEngine or a BatchProcessor."""
    for p, h in zip((ConvolutionPath.Lsl, ConvolutionPath.Lsr, ConvolutionPath.Rsl, ConvolutionPath.Rsr), irs):
        target.set_ir(p, h)
