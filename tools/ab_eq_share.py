#!/usr/bin/env python3
"""EQ || convolution sharing experiment: the headline step at several stream counts, per-kernel times from the
library's HIP-event spans.  Environment knobs are read by the library (OHS_EQ_RING_V1, OHS_EQ_NO_PRIO,
OHS_EQ_WG_WAVES, OHS_NO_OVERLAP, OHS_OVERLAP_CUTS ...): run once per variant, e.g. tools/ab_eq_share.sh.
Usage: python tools/ab_eq_share.py [streams ...]   (default 256 1024 2048)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402

FRAMES = 480256
KNOBS = ("OHS_P1_XCD", "OHS_EQ_XCD", "OHS_XCD_SPLIT", "OHS_FORCE_OVERLAP", "OHS_AB_SIDE_STREAM", "OHS_CONV_CU_MASK", "OHS_EQ_LDS", "OHS_EQ_RING_V1", "OHS_EQ_NO_PRIO", "OHS_EQ_WG_WAVES", "OHS_NO_OVERLAP", "OHS_OVERLAP_CUTS", "OHS_P1_TARGET_WAVES",
         "OHS_EQ_PRIO", "OHS_LIB")


def run(S, steps=4, warmup=3):
    dev = torch.device("cuda:0")
    bp = ohs.BatchProcessor(S, num_bands=10)
    irs = synth.hrir_set(512)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(synth.eq_table()):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    x = synth.white_noise_torch(0, S, FRAMES, dev)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    # OHS_AB_SIDE_STREAM=1: the calls go to a non-blocking stream of their own instead of the legacy default stream
    side = torch.cuda.Stream() if os.environ.get("OHS_AB_SIDE_STREAM") == "1" else None
    hs = side.cuda_stream if side is not None else None
    for _ in range(warmup):
        bp.process(x, out=y, hip_stream=hs)
    torch.cuda.synchronize()
    bp.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        bp.process(x, out=y, hip_stream=hs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    eq_ms, conv_ms, n, _, _ = bp.profile_read()
    eb, cb = bp.kernel_bytes(FRAMES // 512)
    print(json.dumps({"streams": S, "ms_per_step": round(dt * 1e3, 3), "Gsamples_s": round(S * FRAMES / dt / 1e9, 2),
                      "eq_ms": round(eq_ms / n, 3), "conv_ms": round(conv_ms / n, 3),
                      "hbm_frac_design_bytes": round((eb + cb) / dt / 8e12, 4),
                      "env": {k: os.environ[k] for k in KNOBS if k in os.environ}}), flush=True)
    del bp, x, y
    torch.cuda.empty_cache()


if __name__ == "__main__":
    for S in [int(a) for a in sys.argv[1:]] or [256, 1024, 2048]:
        run(S)
