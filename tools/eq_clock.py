#!/usr/bin/env python3
"""Shader clock inside k_eq_ring (experiment build: OHS_BUILD_TAG=eqstamps OHS_EXTRA_DEFS=-DOHS_EQ_STAMPS): s_memtime
(shader clock) against s_memrealtime (100 MHz) over every wave of the last EQ launch of a steady-state headline step,
with and without the convolution running beside it.
    OHS_LIB=open_headstage_amd/libohs_hip_eqstamps.so python tools/eq_clock.py [streams ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import _ffi, synth  # noqa: E402

FRAMES = 480256
dev = torch.device("cuda:0")
L = _ffi.lib()
L.ohs_debug_eq_stamps.argtypes = [C.c_void_p, C.c_size_t]


def run(S, conv_on):
    bp = ohs.BatchProcessor(S, num_bands=10)
    irs = synth.hrir_set(512)
    for p in range(4):
        bp.set_ir(p, irs[p] if conv_on else irs[p][:0])
    for i, b in enumerate(synth.eq_table()):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    x = synth.white_noise_torch(0, S, FRAMES, dev)
    y = torch.empty_like(x)
    for _ in range(4):
        bp.process(x, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        bp.process(x, out=y)
    b.record()
    torch.cuda.synchronize()
    n_waves = min(4096, (2 * S + 3) // 4)
    buf = (C.c_ulonglong * (4 * n_waves))()
    assert L.ohs_debug_eq_stamps(buf, n_waves) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(n_waves, 4).astype(np.int64)
    st = st[st[:, 1] > st[:, 0]]
    us = (st[:, 1] - st[:, 0]) / 100.0
    ghz = (st[:, 3] - st[:, 2]) / ((st[:, 1] - st[:, 0]) * 10.0)
    print(f"streams {S} convolution {'on' if conv_on else 'muted (empty IRs: general path, no P = 1 kernel)'}: "
          f"{a.elapsed_time(b) / 4:.3f} ms per step; last EQ launch: {len(st)} waves, life {us.mean():.1f} us "
          f"(min {us.min():.1f}, max {us.max():.1f}), shader clock {ghz.mean():.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f})",
          flush=True)
    del bp, x, y
    torch.cuda.empty_cache()


for S in [int(a) for a in sys.argv[1:]] or [256, 1024, 2048]:
    run(S, True)
