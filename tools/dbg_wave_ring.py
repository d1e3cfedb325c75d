import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import open_headstage_amd as ohs
from open_headstage_amd import synth, _ffi, dsp
from oracle import ohs_oracle as orc
orc.build()
dsp.lib = _ffi.experiments_lib
_ffi.set_tuning("eq_form", 2)
bands = synth.eq_table()
for n in (192, 240, 288, 400, 1000):
    eg = ohs.StereoParametricEQ.new(len(bands), 48000.0)
    eo = orc.StereoParametricEQ(len(bands), 48000.0)
    for i, b in enumerate(bands):
        eg.update_band_coeffs(i, 48000.0, b)
        c, en = eg.get_band_coeffs(i)
        eo.set_band_coeffs(i, c, en)
    x = synth.white_noise([31], 2 * n)[0]
    for call in range(2):
        gl, gr = x[0, call * n:(call + 1) * n].copy(), x[1, call * n:(call + 1) * n].copy()
        ol, orr = gl.copy(), gr.copy()
        eg.process_block(gl, gr); eo.process_block(ol, orr)
        d = np.nonzero(gl.view(np.uint32) != ol.view(np.uint32))[0]
        print(f"n={n} call {call}: {len(d)} differ", (f"first {d[:6]} last {d[-3:]}; values got {gl[d[:3]]} want {ol[d[:3]]}; input there {x[0, call*n + d[:3]]}" if len(d) else ""))
