"""ohs_batch_process_host (host buffers, copy-in / kernels / copy-out pipelined over time chunks) against the
device-buffer entry point: bit-identical to the same sequence of chunk-sized ohs_batch_process calls, within the
parity bar of one whole-length call and of the oracle; ragged last chunk, in-place host buffers, numpy and
pinned torch buffers, per-stream (irregular) strides."""
import ctypes as C

import numpy as np
import pytest

from tests.util import assert_parity

pytestmark = pytest.mark.gpu


def _make(S, taps=512, eq=True, gain=0.7):
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    bands = synth.eq_table()
    bp = ohs.BatchProcessor(S, num_bands=len(bands))
    irs = synth.hrir_set(taps)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(bands):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(eq)
    bp.set_gain(gain)
    return bp, irs, bands


@pytest.mark.parametrize("S,n_blocks,chunk", [(3, 200, 64), (5, 130, 64), (2, 64, 0), (4, 70, 100), (1, 9, 4)])
@pytest.mark.parametrize("taps", [512, 1300])
def test_host_pipeline_matches_chunked_device_calls_bit_for_bit(S, n_blocks, chunk, taps):
    import torch
    from open_headstage_amd import synth
    frames = n_blocks * 512
    x = synth.white_noise(range(S), frames)
    bp, _, _ = _make(S, taps)
    y_host = bp.process_host(x.copy(), chunk_blocks=chunk)
    # the same chunking through the device entry point
    bp2, _, _ = _make(S, taps)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    # (chunk 0 = the library's default rule: ~32 MiB per direction, at most an eighth of the call, at least 16 blocks)
    cb = chunk if chunk else max(16, min((32 << 20) // (S * 2 * 512 * 4), max(n_blocks // 8, 1)))
    cb = min(cb, n_blocks)
    for b0 in range(0, n_blocks, cb):
        nb = min(cb, n_blocks - b0)
        xi = xd[:, :, b0 * 512:(b0 + nb) * 512].contiguous()
        yd[:, :, b0 * 512:(b0 + nb) * 512] = bp2.process(xi)
    torch.cuda.synchronize()
    assert np.array_equal(y_host, yd.cpu().numpy()), "host pipeline differs from the chunked device calls"
    # and within the parity bar of ONE whole-length device call (the last block of a call keeps four separate
    # overlaps, so the rounding of chunk-boundary blocks differs)
    bp3, _, _ = _make(S, taps)
    y1 = bp3.process(xd).cpu().numpy()
    assert_parity(y_host, y1, "host pipeline vs one device call")


def test_host_pipeline_in_place_pinned_and_state_carries_over():
    import torch
    from open_headstage_amd import synth
    from oracle import ohs_oracle as orc
    import open_headstage_amd as ohs
    S, n_blocks = 4, 96
    frames = n_blocks * 512
    x = synth.white_noise(range(S), 2 * frames)
    bp, irs, bands = _make(S, gain=1.0)
    buf = torch.from_numpy(x[:, :, :frames].copy()).pin_memory()
    bp.process_host(buf, out=buf, chunk_blocks=40)                     # in place, pinned
    first = buf.numpy().copy()
    buf2 = torch.from_numpy(x[:, :, frames:].copy()).pin_memory()
    second = bp.process_host(buf2, chunk_blocks=17).numpy()            # state (EQ, overlaps) continues
    got = np.concatenate([first, second], axis=2)
    eng = orc.ConvolutionEngine()
    for p in range(4):
        eng.set_ir(p, irs[p])
    eq = orc.StereoParametricEQ(len(bands), synth.FS)
    for i, b in enumerate(bands):
        eq.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), True)
    l, r = x[1, 0].copy(), x[1, 1].copy()
    for c in range(0, 2 * frames, 1024):
        orc.chain_process(eng, eq, l[c:c + 1024], r[c:c + 1024], eq_enable=True, gain=1.0)
    assert_parity(got[1], np.stack([l, r]), "host pipeline, two calls, stream 1 vs oracle")


def test_host_pipeline_irregular_strides_and_errors():
    from open_headstage_amd import OhsError, synth
    from open_headstage_amd._ffi import lib
    S, n_blocks = 3, 70
    frames = n_blocks * 512
    x = synth.white_noise(range(S), frames)
    bp, _, _ = _make(S, eq=False)
    ref = bp.process_host(x.copy(), chunk_blocks=32)
    # per-stream rows with padding: channel stride frames + 16, stream stride 2 * channel stride + 64
    cs, ss = frames + 16, 2 * (frames + 16) + 64
    big_in = np.zeros(S * ss, np.float32)
    big_out = np.full(S * ss, 7.0, np.float32)
    for s in range(S):
        for c in range(2):
            big_in[s * ss + c * cs: s * ss + c * cs + frames] = x[s, c]
    bp2, _, _ = _make(S, eq=False)
    rc = lib().ohs_batch_process_host(bp2._h, C.c_void_p(big_in.ctypes.data), C.c_void_p(big_out.ctypes.data), n_blocks,
                                      ss, cs, 32)
    assert rc == 0
    for s in range(S):
        for c in range(2):
            assert np.array_equal(big_out[s * ss + c * cs: s * ss + c * cs + frames], ref[s, c])
            assert np.all(big_out[s * ss + c * cs + frames: s * ss + c * cs + frames + 16] == 7.0)   # padding untouched
    with pytest.raises(OhsError):
        from open_headstage_amd._ffi import check
        check(lib().ohs_batch_process_host(bp2._h, None, C.c_void_p(big_out.ctypes.data), n_blocks, ss, cs, 0))
    with pytest.raises(OhsError):
        check(lib().ohs_batch_process_host(bp2._h, C.c_void_p(big_in.ctypes.data), C.c_void_p(big_out.ctypes.data),
                                           n_blocks, ss, frames - 1, 0))


def test_host_pipeline_stage_trace_diagnostic_runs():
    """The stage trace is a diagnostic of the EXPERIMENTS build (libohs_hip_exp.so, OHS_HOST_PIPE_TRACE=1; the product
    library reads no environment variable): a child process runs the package on that build -- the per-chunk stage table
    is printed and the results are unchanged."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import open_headstage_amd as ohs
from open_headstage_amd import synth
S, nb = 3, 96
x = synth.white_noise(range(S), nb * 512)
def make():
    bp = ohs.BatchProcessor(S, num_bands=10)
    irs = synth.hrir_set(512)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(synth.eq_table()):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    return bp
a = make().process_host(x.copy(), chunk_blocks=32)
b = make().process_host(x.copy(), chunk_blocks=32)
assert np.array_equal(a, b)
print("trace ok")
""" % root
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OHS_HOST_PIPE_TRACE="1", OHS_LIB="exp"),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "trace ok" in res.stdout, (res.stdout[-300:], res.stderr[-1500:])
    assert "[ohs host pipeline] 3 chunks of 32 blocks" in res.stderr, res.stderr[-1500:]
