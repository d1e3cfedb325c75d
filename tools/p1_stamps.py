#!/usr/bin/env python3
"""Per-wave life spans of k_conv_p1 (experiment build: OHS_BUILD_TAG=stamps OHS_EXTRA_DEFS=-DOHS_P1_STAMPS).
Prints, for the last launch of a steady-state run of config 3 without EQ: the spread of wave start / end times, the
time-averaged number of resident waves, and the same per workgroup (16 waves = one CU's complement).
    OHS_BUILD_TAG=stamps OHS_EXTRA_DEFS=-DOHS_P1_STAMPS python -m open_headstage_amd.build
    OHS_LIB=open_headstage_amd/libohs_hip_stamps.so python tools/p1_stamps.py [streams]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import _ffi, synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = 480256
dev = torch.device("cuda:0")
bp = ohs.BatchProcessor(S, num_bands=10)
irs = synth.hrir_set(512)
for p in range(4):
    bp.set_ir(p, irs[p])
bp.set_eq_enabled(False)
x = synth.white_noise_torch(0, S, frames, dev)
y = torch.empty_like(x)
for _ in range(30):
    bp.process(x, out=y)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(8):
    bp.process(x, out=y)
b.record()
torch.cuda.synchronize()
print(f"steady state: {a.elapsed_time(b) / 8:.4f} ms per step")
L = _ffi.lib()
n_waves = int(os.environ.get("P1_WAVES", "8192"))
buf = (C.c_ulonglong * (4 * n_waves))()
L.ohs_debug_p1_stamps.argtypes = [C.c_void_p, C.c_size_t]
rc = L.ohs_debug_p1_stamps(buf, n_waves)
assert rc == 0, rc
st = np.frombuffer(buf, dtype=np.uint64).reshape(n_waves, 4).astype(np.int64)
ok = st[:, 2] > 0
st = st[ok]
xcc = (st[:, 3] >> 32) & 0xF
print("waves per XCC id:", {int(k): int(v) for k, v in zip(*np.unique(xcc, return_counts=True))})
t0 = st[:, 0].min()
entry, start, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0, (st[:, 2] - t0) / 100.0      # microseconds
print(f"waves {len(st)}; kernel span {end.max():.1f} us; wave life (start->end) mean {np.mean(end - start):.1f} us, "
      f"min {np.min(end - start):.1f}, max {np.max(end - start):.1f}; table fill + barrier mean {np.mean(start - entry):.1f} us")
span = end.max()
print(f"time-averaged resident computing waves: {np.sum(end - start) / span:.0f} of 4096 slots "
      f"({np.sum(end - start) / span / 4096:.1%})")
# per workgroup (consecutive 16 waves)
g = len(st) // 16
wg_entry = entry[:g * 16].reshape(g, 16).min(1)
wg_end = end[:g * 16].reshape(g, 16).max(1)
life = (end - start)[:g * 16].reshape(g, 16)
print(f"workgroups {g}: span mean {np.mean(wg_end - wg_entry):.1f} us; inside a workgroup the waves' lives spread "
      f"min/max = {np.mean(life.min(1)):.1f} / {np.mean(life.max(1)):.1f} us (mean of workgroups), "
      f"slot use inside a workgroup {np.mean(life.sum(1) / (16 * (wg_end - wg_entry))):.1%}")
order = np.argsort(wg_entry)
first, second = order[:256], order[256:]
if len(second) == 0:
    second = first
print(f"first round: entry {wg_entry[first].min():.1f}..{wg_entry[first].max():.1f} us, end {wg_end[first].min():.1f}.."
      f"{wg_end[first].max():.1f}; second round: entry {wg_entry[second].min():.1f}..{wg_entry[second].max():.1f}, "
      f"end {wg_end[second].min():.1f}..{wg_end[second].max():.1f}")
# by wave index inside the workgroup: do older waves finish earlier?
print("mean life by wave index in workgroup:", np.round(life.mean(0), 1))
hist, edges = np.histogram(end, bins=12, range=(0, span))
print("wave ends per 1/12 of the span:", hist)

try:
    buf2 = (C.c_ulonglong * (2 * n_waves))()
    L.ohs_debug_p1_clk.argtypes = [C.c_void_p, C.c_size_t]
    if L.ohs_debug_p1_clk(buf2, n_waves) == 0:
        ck = np.frombuffer(buf2, dtype=np.uint64).reshape(n_waves, 2).astype(np.int64)[ok]
        ghz = (ck[:, 1] - ck[:, 0]) / ((st[:, 2] - st[:, 1]) * 10.0) / 1e0 * 1e-3 * 1e3 / 1e3      # cycles per 10 ns -> GHz
        ghz = (ck[:, 1] - ck[:, 0]) / ((st[:, 2] - st[:, 1]) * 10.0)
        gi = ghz[:g * 16].reshape(g, 16)
        print("in-kernel shader clock (GHz) by wave index in workgroup:", np.round(gi.mean(0), 3), f"overall {ghz.mean():.3f}")
except AttributeError:
    pass
