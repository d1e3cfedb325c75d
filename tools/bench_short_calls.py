#!/usr/bin/env python3
"""Prices short batch calls on long impulse responses: BASELINE config-4 shape -- 64 streams,
4 x 16 384 taps -- driven with calls of 1, 2, 3, 4, 8, 32 and 256 blocks of 512 frames.  Device time per call by HIP
events around a run of back-to-back calls on one stream (so launch gaps between a call's kernels count, host
enqueue time does not unless it is the bound), and host wall time per call beside it.  One JSON line per call size.

    python tools/bench_short_calls.py [--streams 64] [--taps 16384] [--plan 0]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--taps", type=int, default=16384)
    ap.add_argument("--plan", type=int, default=0)
    ap.add_argument("--sizes", type=str, default="1,2,3,4,8,32,256")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    S = a.streams
    bp = ohs.BatchProcessor(S, num_bands=10)
    irs = synth.hrir_set(a.taps)
    for p in range(4):
        bp.set_ir(p, irs[p])
    bp.set_eq_enabled(False)
    bp.set_conv_plan(a.plan)
    frames_total = 256 * 512
    x = synth.white_noise_torch(0, S, frames_total, dev)
    y = torch.empty_like(x)
    # a long call first: every path is then older than its own impulse response (the steady state of a running job)
    bp.process(x, out=y)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream(dev)
    steady = None
    for nb in [int(v) for v in a.sizes.split(",")]:
        n = nb * 512
        calls = max(8, min(400, 4096 // nb))
        xs = x[:, :, :n].contiguous()
        ys = torch.empty_like(xs)
        for _ in range(3):
            bp.process(xs, out=ys)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st)
        for _ in range(calls):
            bp.process(xs, out=ys)
        e1.record(st)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_wall = time.perf_counter() - t0
        dev_us = e0.elapsed_time(e1) * 1e3 / calls
        rec = {"streams": S, "taps": a.taps, "plan": a.plan, "blocks_per_call": nb, "calls": calls,
               "device_us_per_call": round(dev_us, 2), "host_enqueue_us_per_call": round(t_host * 1e6 / calls, 2),
               "wall_us_per_call": round(t_wall * 1e6 / calls, 2),
               "us_per_block": round(dev_us / nb, 3)}
        if nb == 256:
            steady = dev_us
        print(json.dumps(rec), flush=True)
    if steady:
        print(json.dumps({"steady_256_block_call_us": round(steady, 2), "per_2_blocks_us": round(steady / 128, 3),
                          "target_3x_us": round(3 * steady / 128, 3)}), flush=True)


if __name__ == "__main__":
    main()
