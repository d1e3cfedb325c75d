#!/usr/bin/env python3
"""Times the other BASELINE.json configs (not the headline bench): C2 single stream, C4 long IR.
Prints one JSON line per config.  Usage: python tools/bench_configs.py [c2] [c4] [c3noeq] [c4engine]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402


def run_batch(name, S, taps, frames, eq_on, steps=5, warmup=2):
    dev = torch.device("cuda:0")
    bp = ohs.BatchProcessor(S, num_bands=10)
    irs = synth.hrir_set(taps)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(synth.eq_table()):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(eq_on)
    x = synth.white_noise_torch(0, S, frames, dev)
    y = torch.empty_like(x)
    for _ in range(warmup):
        bp.process(x, out=y)
    torch.cuda.synchronize()
    bp.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        bp.process(x, out=y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    eq_ms, conv_ms, n, _, _ = bp.profile_read()
    eb, cb = bp.kernel_bytes(frames // 512)
    print(json.dumps({"config": name, "streams": S, "taps": taps, "frames": frames, "eq": eq_on,
                      "ms_per_step": round(dt * 1e3, 3), "Msamples_per_s": round(S * frames / dt / 1e6, 1),
                      "x_realtime": round(S * frames / dt / 48000.0, 0),
                      "eq_ms": round(eq_ms / n, 3), "conv_ms": round(conv_ms / n, 3),
                      "conv_model_GBps": round(cb / (conv_ms / n * 1e-3) / 1e9, 1)}), flush=True)


def _per_call(fn, items, gap_us=0.0):
    """gap_us > 0: the host idles that long in front of every call (not timed) -- a real-time host's calls are 21 ms apart,
    and what the engine kernels compute ahead behind a call's completion word is then done when the next call arrives"""
    import gc
    gc.collect()
    gc.disable()        # a full collection with torch loaded takes ~40 ms and would land on one call
    ts = []
    for it in items:
        if gap_us > 0:
            tw = time.perf_counter() + gap_us * 1e-6
            while time.perf_counter() < tw:
                pass
        t0 = time.perf_counter()
        fn(*it)
        ts.append(time.perf_counter() - t0)
    gc.enable()
    ts = np.array(ts) * 1e6
    return {"median": round(float(np.median(ts)), 1), "mean": round(float(ts.mean()), 1),
            "p99": round(float(np.percentile(ts, 99)), 1), "max": round(float(ts.max()), 1),
            "argmax": int(ts.argmax())}


def run_engine_c2(calls=400, n=1024):
    """C2: one stream, 1024-frame host blocks through the blocking C ABI (host buffers in, host buffers
    out), per-call latency in microseconds: the convolution engine alone, the EQ alone, and the whole
    lib.rs:1169-1207 chain (coefficient refresh excluded)."""
    e = ohs.ConvolutionEngine.new()
    irs = synth.hrir_set(512)
    for p in range(4):
        e.set_ir(p, irs[p])
    bands = synth.eq_table()
    eq = ohs.StereoParametricEQ.new(len(bands), synth.FS)
    for i, b in enumerate(bands):
        eq.update_band_coeffs(i, synth.FS, b)
    x = synth.white_noise([0], n * (calls + 20))[0]
    ol = np.empty(n, np.float32); orr = np.empty(n, np.float32)

    def blocks():
        return [(x[0, c * n:(c + 1) * n].copy(), x[1, c * n:(c + 1) * n].copy()) for c in range(calls + 20)]

    bl = blocks()
    for l, r in bl[:20]:
        e.process_block(l, r, ol, orr)
    st = _per_call(lambda l, r: e.process_block(l, r, ol, orr), bl[20:])
    print(json.dumps({"config": f"c2_engine_{n}_blocking", "us_per_call": st["median"], "latency_us": st,
                      "x_realtime": round(n / (st["median"] * 1e-6) / 48000.0, 0)}), flush=True)
    # the same calls served by the resident kernel (ohs_engine_set_realtime): no launch, no stream sync
    e.set_realtime(True)
    bl = blocks()
    for l, r in bl[:20]:
        e.process_block(l, r, ol, orr)
    st = _per_call(lambda l, r: e.process_block(l, r, ol, orr), bl[20:])
    print(json.dumps({"config": f"c2_engine_{n}_realtime_resident_kernel", "us_per_call": st["median"], "latency_us": st,
                      "x_realtime": round(n / (st["median"] * 1e-6) / 48000.0, 0)}), flush=True)
    # protocol floor: 8-frame calls (64 of them fill one block), i.e. mailbox round trips with almost no work
    e2 = ohs.ConvolutionEngine.new()
    for p in range(4):
        e2.set_ir(p, irs[p])
    e2.set_realtime(True)
    tiny = [(np.zeros(8, np.float32), np.zeros(8, np.float32)) for _ in range(400)]
    o8 = np.empty(8, np.float32); o8b = np.empty(8, np.float32)
    for l, r in tiny[:20]:
        e2.process_block(l, r, o8, o8b)
    st = _per_call(lambda l, r: e2.process_block(l, r, o8, o8b), tiny[20:])
    print(json.dumps({"config": "c2_resident_mailbox_round_trip_8_frames", "us_per_call": st["median"], "latency_us": st}), flush=True)
    e2.set_realtime(False)
    e.set_realtime(False)
    bl = blocks()
    for l, r in bl[:20]:
        eq.process_block(l, r)
    st = _per_call(lambda l, r: eq.process_block(l, r), bl[20:])
    print(json.dumps({"config": f"c2_eq_only_{n}_blocking", "us_per_call": st["median"], "latency_us": st}), flush=True)
    bl = blocks()
    for l, r in bl[:20]:
        ohs.process_chain(e, eq, l, r, eq_enable=True, output_gain=0.9)
    st = _per_call(lambda l, r: ohs.process_chain(e, eq, l, r, eq_enable=True, output_gain=0.9), bl[20:])
    print(json.dumps({"config": f"c2_chain_eq_conv_gain_{n}_blocking", "us_per_call": st["median"], "latency_us": st,
                      "x_realtime": round(n / (st["median"] * 1e-6) / 48000.0, 0)}), flush=True)


def run_engine_long_ir(taps_list=(2048, 4096, 8192, 16384), calls=300, n=1024):
    """The plugin's own use of config 4's response: ONE engine, a room response of `taps` taps per path, 1 024-frame
    host blocks through the blocking C ABI (launch per call, then the resident kernel)."""
    for taps in taps_list:
        e = ohs.ConvolutionEngine.new()
        irs = synth.hrir_set(taps)
        for p in range(4):
            e.set_ir(p, irs[p])
        x = synth.white_noise([0], n * (calls + 40))[0]
        ol = np.empty(n, np.float32); orr = np.empty(n, np.float32)
        bl = [(x[0, c * n:(c + 1) * n].copy(), x[1, c * n:(c + 1) * n].copy()) for c in range(calls + 40)]
        for mode in ("launch_per_call", "resident_kernel"):
            e.set_realtime(mode == "resident_kernel")
            for l, r in bl[:40]:
                e.process_block(l, r, ol, orr)
            st = _per_call(lambda l, r: e.process_block(l, r, ol, orr), bl[40:])
            paced = _per_call(lambda l, r: e.process_block(l, r, ol, orr), bl[40:], gap_us=300.0)
            print(json.dumps({"config": f"c4_engine_{n}_{mode}", "taps": taps, "partitions": (taps + 511) // 512,
                              "us_per_call_back_to_back": st["median"], "us_per_call_300us_apart": paced["median"],
                              "latency_us_back_to_back": st, "latency_us_300us_apart": paced,
                              "x_realtime_back_to_back": round(n / (st["median"] * 1e-6) / 48000.0, 0)}), flush=True)
        e.set_realtime(False)
        del e


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c4", "c3noeq", "c4engine"]
    if "c2" in which:
        run_engine_c2()
        run_batch("c2_batch_1stream", 1, 512, 480256, False)
    if "c3noeq" in which:
        # 30 untimed steps: the first full-chip steps after start run ~25 % slower than the steady state (bench.py, conv_alone)
        run_batch("c3_noeq", 256, 512, 480256, False, steps=16, warmup=30)
    if "c4engine" in which:
        run_engine_long_ir()
    if "c4" in which:
        run_batch("c4_p32_64streams", 64, 16384, 512 * 256, False, steps=16, warmup=30)
