"""AutoEQ profile ingestion -- host-side mirror of src/autoeq_parser.rs.

`parse_autoeq_csv(path)` returns the reference's `Vec<BandSetting>`; `apply_bands` pushes them into a
`StereoParametricEQ` or a `BatchProcessor` the way the plugin's editor applies a loaded profile
(one band per row, enabled)."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

from ._ffi import BandSettingC, check, lib
from .dsp import BandConfig, FilterType


@dataclass
class BandSetting:                     # autoeq_parser.rs:34-41
    enabled: bool = False
    filter_type: FilterType = FilterType.Peak
    frequency: float = 0.0
    q: float = 0.0
    gain: float = 0.0


def _collect(call) -> list[BandSetting]:
    n = C.c_size_t()
    check(call(None, 0, C.byref(n)))
    arr = (BandSettingC * max(1, n.value))()
    check(call(arr, n.value, C.byref(n)))
    return [BandSetting(bool(a.enabled), FilterType(a.filter_type), a.frequency, a.q, a.gain)
            for a in arr[: n.value]]


def parse_autoeq_csv(path) -> list[BandSetting]:
    p = os.fsencode(path)
    return _collect(lambda out, cap, cnt: lib().ohs_autoeq_parse_csv(p, out, cap, cnt))


def parse_autoeq_csv_text(text: str | bytes) -> list[BandSetting]:
    t = text.encode() if isinstance(text, str) else text
    return _collect(lambda out, cap, cnt: lib().ohs_autoeq_parse_csv_text(t, len(t), out, cap, cnt))


def apply_bands(target, bands: list[BandSetting], sample_rate: float) -> None:
    """target: StereoParametricEQ or BatchProcessor; rows beyond its band count are ignored
    (update_band_coeffs ignores idx >= num_bands, parametric_eq.rs:145)."""
    for i, b in enumerate(bands):
        target.update_band_coeffs(i, sample_rate, BandConfig(b.filter_type, b.frequency, b.q, b.gain, b.enabled))
