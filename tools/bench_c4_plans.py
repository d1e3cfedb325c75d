#!/usr/bin/env python3
"""BASELINE config 4 (64 streams, 4 x 16 384 taps, 256-block calls) under the block-512 and the block-2048 plan: ms per step
(HIP events around K back-to-back calls on one stream), Msamples/s, and the plan the library reports.
    python tools/bench_c4_plans.py [--streams 64] [--taps 16384] [--blocks 256] [--steps 20]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--taps", type=int, default=16384)
    ap.add_argument("--blocks", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--plans", type=str, default="1,2,0")
    ap.add_argument("--tune", type=str, default="", help="experiments library, key=value[,key=value...] (e.g. lb_skip=2)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    S, n = a.streams, a.blocks * 512
    irs = synth.hrir_set(a.taps)
    x = synth.white_noise_torch(0, S, n, dev)
    y = torch.empty_like(x)
    st = torch.cuda.current_stream(dev)
    library = None
    if a.tune:
        from open_headstage_amd import _ffi
        library = _ffi.experiments_lib()
        for kv in a.tune.split(","):
            k, v = kv.split("=")
            _ffi.set_tuning(k, v)
    for plan in [int(v) for v in a.plans.split(",")]:
        bp = ohs.BatchProcessor(S, num_bands=10, library=library)
        for p in range(4):
            bp.set_ir(p, irs[p])
        bp.set_eq_enabled(False)
        bp.set_conv_plan(plan)
        for _ in range(3):
            bp.process(x, out=y)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            bp.process(x, out=y)
            e1.record(st)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        med = ts[len(ts) // 2]
        kern, ranges = bp.last_conv_plan()
        _, cb = bp.kernel_bytes(a.blocks)
        print(json.dumps({"tune": a.tune, "streams": S, "taps": a.taps, "blocks": a.blocks, "plan": plan, "kernel": kern, "ranges": ranges,
                          "ms_median": round(med, 4), "ms_min": round(ts[0], 4), "ms_max": round(ts[-1], 4),
                          "Gsamples_per_s": round(S * n / med / 1e6, 2), "design_bytes_MB": round(cb / 1e6, 1),
                          "design_TBps": round(cb / med / 1e9, 3)}), flush=True)


if __name__ == "__main__":
    main()
