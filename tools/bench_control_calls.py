#!/usr/bin/env python3
"""What the calls AROUND process_block cost a host (milliseconds, median of a few): engine create / destroy, set_ir at 512 and
16 384 taps (the reference does P forward transforms there, convolution.rs:120-132), clone (convolution.rs:68), switching
real-time mode, the first call after each, EQ create / update_band_coeffs / reset.  One engine, blocking C ABI."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import synth  # noqa: E402


def med(fn, reps=9, setup=None):
    ts = []
    for _ in range(reps):
        arg = setup() if setup else None
        t0 = time.perf_counter()
        fn(arg) if setup else fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return round(ts[len(ts) // 2] * 1e3, 3)


def main():
    out = {}
    ohs.ConvolutionEngine.new()         # (device context, twiddle tables: once per process)
    engines = []
    out["engine_create_ms"] = med(lambda: engines.append(ohs.ConvolutionEngine.new()))
    e = engines[0]
    x = synth.white_noise([1], 1024)[0]
    l, r = x[0].copy(), x[1].copy()
    ol, orr = np.empty(1024, np.float32), np.empty(1024, np.float32)
    for taps in (512, 16384):
        irs = synth.hrir_set(taps)
        out[f"set_ir_{taps}_taps_ms"] = med(lambda: e.set_ir(1, irs[1]))
        for p in range(4):
            e.set_ir(p, irs[p])
        e.process_block(l, r, ol, orr)
        out[f"first_call_after_set_ir_{taps}_taps_ms"] = med(lambda _: e.process_block(l, r, ol, orr), setup=lambda: e.set_ir(2, irs[2]))
        out[f"clone_{taps}_taps_ms"] = med(lambda: engines.append(e.clone()))
        e.set_realtime(True)
        e.process_block(l, r, ol, orr)
        out[f"set_ir_{taps}_taps_in_realtime_mode_ms"] = med(lambda: e.set_ir(1, irs[1]))
        out[f"first_call_after_set_ir_{taps}_taps_in_realtime_mode_ms"] = med(lambda _: e.process_block(l, r, ol, orr),
                                                                             setup=lambda: e.set_ir(2, irs[2]))
        e.set_realtime(False)
    out["set_realtime_on_plus_first_call_ms"] = med(lambda _: (e.set_realtime(True), e.process_block(l, r, ol, orr)),
                                                    setup=lambda: e.set_realtime(False))
    out["set_realtime_off_ms"] = med(lambda _: e.set_realtime(False), setup=lambda: (e.set_realtime(True), e.process_block(l, r, ol, orr)))
    bands = synth.eq_table()
    eqs = []
    out["eq_create_ms"] = med(lambda: eqs.append(ohs.StereoParametricEQ.new(10, synth.FS)))
    q = eqs[0]
    out["eq_update_10_bands_ms"] = med(lambda: [q.update_band_coeffs(i, synth.FS, b) for i, b in enumerate(bands)])
    out["eq_reset_ms"] = med(lambda: q.reset_all_bands_state())
    t0 = time.perf_counter()
    del engines[:]
    del e
    out["destroy_all_engines_ms_total"] = round((time.perf_counter() - t0) * 1e3, 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
