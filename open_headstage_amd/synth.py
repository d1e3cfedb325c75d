"""Deterministic synthetic inputs: portable splitmix64 PRNG, so the
numpy (CPU) and torch (GPU) generators produce identical bits.

  audio : uniform(-1,1) f32 white noise, seed 0x0A5EAD00 + stream_id (L), + 2^32 (R)
  HRIRs : h[k] = g (delta[k-d] + 0.5 N(0,1) exp(-k/tau)), tau = taps/6,
          (d,g): Lsl (30,1.0) Lsr (45,0.4) Rsl (45,0.4) Rsr (30,1.0); seeds 1..4;
          scaled so sum|h| = 1 per ear pair
  EQ    : the fixed 10-band table of section 8d, fs = 48 kHz
"""
from __future__ import annotations

import numpy as np

from .dsp import BandConfig, FilterType

AUDIO_SEED = 0x0A5EAD00
_GAMMA = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_MASK = (1 << 64) - 1


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n successive splitmix64 outputs starting `offset` draws into the sequence of `seed`."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = (np.uint64(seed & _MASK) + i * np.uint64(_GAMMA))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_pm1(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """uniform in [-1, 1) with a 24-bit mantissa, float32"""
    u = (splitmix64(seed, n, offset) >> np.uint64(40)).astype(np.float32)   # < 2^24: exact
    return u * np.float32(2.0 ** -23) - np.float32(1.0)


def white_noise(stream_ids, frames: int, offset: int = 0) -> np.ndarray:
    """[len(stream_ids), 2, frames] float32"""
    ids = list(stream_ids)
    out = np.empty((len(ids), 2, frames), np.float32)
    for k, sid in enumerate(ids):
        out[k, 0] = uniform_pm1(AUDIO_SEED + sid, frames, offset)
        out[k, 1] = uniform_pm1(AUDIO_SEED + sid + (1 << 32), frames, offset)
    return out


def white_noise_torch(stream_id_start: int, n_streams: int, frames: int, device, offset: int = 0):
    """Same bits as white_noise(), generated on `device` with int64 wrap-around arithmetic."""
    import torch

    def s64(v: int) -> int:   # two's-complement view of a u64 constant
        v &= _MASK
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(z, k):            # logical shift right on int64
        return (z >> k) & ((1 << (64 - k)) - 1)

    sid = torch.arange(stream_id_start, stream_id_start + n_streams, device=device, dtype=torch.int64)
    seeds = torch.stack([sid + AUDIO_SEED, sid + AUDIO_SEED + (1 << 32)], dim=1)        # [S, 2]
    out = torch.empty((n_streams, 2, frames), device=device, dtype=torch.float32)
    step = 1 << 22
    for f0 in range(0, frames, step):
        f1 = min(frames, f0 + step)
        i = torch.arange(offset + f0 + 1, offset + f1 + 1, device=device, dtype=torch.int64)
        z = seeds[:, :, None] + i[None, None, :] * s64(_GAMMA)
        z = (z ^ lsr(z, 30)) * s64(_M1)
        z = (z ^ lsr(z, 27)) * s64(_M2)
        z = z ^ lsr(z, 31)
        u = lsr(z, 40).to(torch.float32)
        out[:, :, f0:f1] = u * (2.0 ** -23) - 1.0
    return out


def _normal(seed: int, n: int) -> np.ndarray:
    """Box-Muller on splitmix64 uniforms (float64)."""
    z = splitmix64(seed, 2 * n)
    u1 = ((z[0::2] >> np.uint64(11)).astype(np.float64) + 1.0) * 2.0 ** -53    # (0, 1]
    u2 = (z[1::2] >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


_PATHS = [(30, 1.0), (45, 0.4), (45, 0.4), (30, 1.0)]     # Lsl, Lsr, Rsl, Rsr : (delay, gain)


def hrir_set(taps: int) -> list[np.ndarray]:
    """[lsl, lsr, rsl, rsr], each float32[taps]."""
    tau = taps / 6.0
    k = np.arange(taps, dtype=np.float64)
    hs = []
    for p, (d, g) in enumerate(_PATHS):
        h = 0.5 * _normal(p + 1, taps) * np.exp(-k / tau)
        if d < taps:
            h[d] += 1.0
        hs.append(g * h)
    # out_l = lsl + rsl ; out_r = lsr + rsr : L1-normalise per ear pair
    nl = np.abs(hs[0]).sum() + np.abs(hs[2]).sum()
    nr = np.abs(hs[1]).sum() + np.abs(hs[3]).sum()
    return [(hs[0] / nl).astype(np.float32), (hs[1] / nr).astype(np.float32),
            (hs[2] / nl).astype(np.float32), (hs[3] / nr).astype(np.float32)]


FS = 48000.0


def eq_table() -> list[BandConfig]:
    bands = [BandConfig(FilterType.LowShelf, 105.0, 0.70, 4.0, True)]
    fcs = [63.0, 125.0, 250.0, 500.0, 1000.0, 2000.0, 4000.0, 8000.0]
    gains = [-3.0, 2.0, -2.5, 1.5, -1.0, 3.0, -4.0, 2.0]
    bands += [BandConfig(FilterType.Peak, f, 1.41, g, True) for f, g in zip(fcs, gains)]
    bands.append(BandConfig(FilterType.HighShelf, 10000.0, 0.70, -3.0, True))
    return bands


def eq_stress_band() -> BandConfig:
    return BandConfig(FilterType.Peak, 20.0, 10.0, 16.0, True)
