#!/usr/bin/env python3
"""One engine through the blocking C ABI, host blocks of 1 024 ... 262 144 frames, responses of 512 and 16 384 taps:
microseconds per call and frames per second -- which call size serves an offline render through process_block best."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402

for taps in (512, 16384):
    irs = synth.hrir_set(taps)
    for n in (1024, 4096, 8192, 16384, 65536, 262144):
        e = ohs.ConvolutionEngine.new()
        for p in range(4):
            e.set_ir(p, irs[p])
        x = synth.white_noise([3], n)[0]
        l, r = x[0].copy(), x[1].copy()
        ol, orr = np.empty(n, np.float32), np.empty(n, np.float32)
        reps = max(6, min(200, (1 << 21) // n))
        for _ in range(4):
            e.process_block(l, r, ol, orr)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            e.process_block(l, r, ol, orr)
            ts.append(time.perf_counter() - t0)
        med = sorted(ts)[len(ts) // 2]
        print(json.dumps({"taps": taps, "frames_per_call": n, "us_per_call": round(med * 1e6, 1),
                          "Mframes_per_s": round(n / med / 1e6, 2), "x_realtime": round(n / med / 48000.0)}), flush=True)
        del e
