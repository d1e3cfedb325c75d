#!/usr/bin/env python3
"""k_eq_ring's two forms (Tuning::eq_form: 1 = four chains per wave, 2 = one chain per wave) on the headline job at several stream
counts: ms per step of the whole batch call (EQ || convolution over six time chunks), experiments library.  Where does one
chain per wave stop paying -- it needs a SIMD per chain, and the convolution underneath needs issue slots too."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import open_headstage_amd as ohs  # noqa: E402
from open_headstage_amd import _ffi, synth  # noqa: E402

L = _ffi.experiments_lib()
frames = 480256
dev = torch.device("cuda:0")
for S in [int(a) for a in sys.argv[1:]] or [128, 256, 320, 384, 448, 512]:
    x = synth.white_noise_torch(0, S, frames, dev)
    y = torch.empty_like(x)
    row = {"streams": S}
    for form in (1, 2, 1, 2):
        _ffi.set_tuning("eq_form", form)
        bp = ohs.BatchProcessor(S, num_bands=10, library=L)
        for p, h in enumerate(synth.hrir_set(512)):
            bp.set_ir(p, h)
        for i, b in enumerate(synth.eq_table()):
            bp.update_band_coeffs(i, synth.FS, b)
        bp.set_eq_enabled(True)
        for _ in range(3):
            bp.process(x, out=y)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
        for i in range(8):
            ev[i].record()
            bp.process(x, out=y)
        ev[8].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(8))
        row.setdefault(f"form_{form}_ms", []).append(round(ts[4], 4))
        del bp
    _ffi.set_tuning("eq_form", 0)
    print(json.dumps(row), flush=True)
    del x, y
    torch.cuda.empty_cache()
