"""Resident real-time mode: ConvolutionEngine::process_block served by a kernel that
stays on the GPU and is fed through a mailbox in pinned host memory.  Same bits as the launch-per-call path and
the oracle's FIFO semantics (ragged host blocks, zero fill), restarts after set_ir / clone / oversize calls /
idle exit, and the latency report of the FIFO adapter."""
import os
import time

import numpy as np
import pytest

from tests.util import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ohs():
    import open_headstage_amd
    return open_headstage_amd


@pytest.fixture()
def synth():
    from open_headstage_amd import synth as s
    return s


def _pair(ohs, synth, taps):
    irs = synth.hrir_set(taps)
    a, b = ohs.ConvolutionEngine.new(), ohs.ConvolutionEngine.new()
    for p in range(4):
        a.set_ir(p, irs[p]); b.set_ir(p, irs[p])
    return a, b, irs


@pytest.mark.parametrize("taps", [128, 512, 1300, 4096])
def test_realtime_mode_is_bit_identical_to_launch_per_call(ohs, synth, taps):
    rt, ref, _ = _pair(ohs, synth, taps)
    rt.set_realtime(True)
    sizes = [1024, 1024, 256, 100, 512, 37, 2048, 1024, 4096, 1, 511, 513, 1024, 8192, 64, 64, 1024]
    x = synth.white_noise([5], sum(sizes))[0]
    o = 0
    for n in sizes:
        l, r = x[0, o:o + n].copy(), x[1, o:o + n].copy()
        gl, gr = rt.process_block(l, r)
        el, er = ref.process_block(l, r)
        assert np.array_equal(gl.view(np.uint32), el.view(np.uint32)) and np.array_equal(gr.view(np.uint32), er.view(np.uint32)), (taps, n, o)
        o += n
    assert rt.silent_frames() == ref.silent_frames()


def test_realtime_mode_against_the_oracle_with_restarts(ohs, synth, oracle):
    rt, _, irs = _pair(ohs, synth, 512)
    eo = oracle.ConvolutionEngine()
    for p in range(4):
        eo.set_ir(p, irs[p])
    rt.set_realtime(True)
    x = synth.white_noise([6], 60000)[0]
    outs_g, outs_o = [], []
    o = 0
    rng = np.random.default_rng(3)
    step = 0
    while o + 20000 < x.shape[1]:
        n = int(rng.choice([1024, 1024, 1024, 480, 256, 2048, 10000]))     # 10000 > 8192: not served by the kernel
        l, r = x[0, o:o + n].copy(), x[1, o:o + n].copy()
        outs_g.append(np.stack(rt.process_block(l, r)))
        outs_o.append(np.stack(eo.process_block(l, r)))
        o += n
        step += 1
        if step == 7:                       # set_ir stops the kernel; the next call restarts it
            irs2 = synth.hrir_set(700)
            rt.set_ir(1, irs2[1]); eo.set_ir(1, irs2[1])
        if step == 11:                      # a clone taken while the kernel is resident carries the state
            c = rt.clone()
            co = eo.clone()
            l2, r2 = x[0, o:o + 1024].copy(), x[1, o:o + 1024].copy()
            assert_parity(np.stack(c.process_block(l2, r2)), np.stack(co.process_block(l2, r2)), "clone of a real-time engine")
    assert_parity(np.concatenate(outs_g, 1), np.concatenate(outs_o, 1), "real-time engine vs oracle")
    rt.set_realtime(False)
    l, r = x[0, o:o + 1024].copy(), x[1, o:o + 1024].copy()
    assert_parity(np.stack(rt.process_block(l, r)), np.stack(eo.process_block(l, r)), "after leaving real-time mode")


def test_resident_kernel_leaves_by_itself_when_idle_and_is_restarted(ohs, synth):
    """ohs_rt_configure(idle_polls) is process-wide, so the idle exit is exercised in a child process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, time, numpy as np
sys.path.insert(0, %r)
import open_headstage_amd as ohs
from open_headstage_amd import synth, _ffi
_ffi.check(_ffi.lib().ohs_rt_configure(-1, 20000))       # ~30 ms at ~1.5 us per poll
irs = synth.hrir_set(512)
a, b = ohs.ConvolutionEngine.new(), ohs.ConvolutionEngine.new()
for p in range(4):
    a.set_ir(p, irs[p]); b.set_ir(p, irs[p])
a.set_realtime(True)
x = synth.white_noise([9], 8 * 1024)[0]
for i in range(8):
    if i in (3, 6):
        time.sleep(0.6)            # > idle limit: the kernel has left, this call must restart it
    l, r = x[0, i*1024:(i+1)*1024].copy(), x[1, i*1024:(i+1)*1024].copy()
    g = a.process_block(l, r); e = b.process_block(l, r)
    assert np.array_equal(g[0], e[0]) and np.array_equal(g[1], e[1]), i
print("idle-exit ok")
""" % root
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "idle-exit ok" in res.stdout, (res.stdout[-500:], res.stderr[-2000:])


def test_resident_kernels_are_capped_per_device_and_the_rest_falls_back(ohs, synth):
    """A resident kernel holds a CU for good, so their number per device is capped (ohs_rt_configure, process-wide ->
    child process): engines beyond the cap run launch per call with the same bits, and get a resident kernel once
    another engine leaves real-time mode."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, %r)
import open_headstage_amd as ohs
from open_headstage_amd import synth, _ffi
_ffi.check(_ffi.lib().ohs_rt_configure(2, 0))
irs = synth.hrir_set(512)
rt = [ohs.ConvolutionEngine.new() for _ in range(4)]
ref = [ohs.ConvolutionEngine.new() for _ in range(4)]
for e in rt + ref:
    for p in range(4):
        e.set_ir(p, irs[p])
for e in rt:
    e.set_realtime(True)
def resident(e):        # has a resident kernel ever served this engine?
    st = (C.c_uint64 * 16)()
    return _ffi.lib().ohs_engine_debug_rt_stamps(e._h, st) == 0
x = synth.white_noise([21], 12 * 1024)[0]
def step(i):
    for k in range(4):
        l, r = x[0, i*1024:(i+1)*1024].copy() * (k + 1) / 4, x[1, i*1024:(i+1)*1024].copy()
        g = rt[k].process_block(l, r); w = ref[k].process_block(l, r)
        assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]), (i, k)
for i in range(4):
    step(i)
assert [resident(e) for e in rt] == [True, True, False, False], [resident(e) for e in rt]
rt[0].set_realtime(False)            # frees a slot
for i in range(4, 8):
    step(i)
assert resident(rt[2]) and not resident(rt[3]), [resident(e) for e in rt]
del rt[1]; del ref[1]                # destroying an engine frees its slot too
import gc; gc.collect()
for i in range(8, 12):
    for k in range(3):
        l, r = x[0, i*1024:(i+1)*1024].copy(), x[1, i*1024:(i+1)*1024].copy()
        g = rt[k].process_block(l, r); w = ref[k].process_block(l, r)
        assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]), (i, k)
assert resident(rt[2])               # the former rt[3]
print("cap ok")
""" % root
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "cap ok" in res.stdout, (res.stdout[-500:], res.stderr[-2000:])


def test_latency_report_matches_the_fifo_rule(ohs, synth, oracle):
    e = ohs.ConvolutionEngine.new()
    assert e.latency_samples(512) == 0 and e.latency_samples(1024) == 0 and e.latency_samples(4096) == 0
    assert e.latency_samples(256) == 256                      # one silent call, then every call is served
    for n in (1, 37, 100, 256, 300, 480, 511, 513, 700, 1000, 1025, 3000):
        # the rule itself, straight from convolution.rs:149-181, run long enough to settle
        in_len = out_len = silent = 0
        for _ in range(6000):
            in_len += n
            k = in_len // 512
            in_len -= 512 * k
            out_len += 512 * k
            if out_len >= n:
                out_len -= n
            else:
                silent += n
        assert e.latency_samples(n) == silent, n
    # and the engine's own counter follows it on real calls (delta HRIR: the served calls, put end to end, are
    # the input stream; the silent calls -- not necessarily the first ones -- are the inserted delay)
    eng = ohs.ConvolutionEngine.new()
    d = np.zeros(8, np.float32); d[0] = 1.0
    eng.set_ir(0, d); eng.set_ir(3, d)
    n = 300
    x = synth.white_noise([2], 60 * n)[0]
    served, in_len, out_len, silent = [], 0, 0, 0
    for c in range(60):
        g = np.stack(eng.process_block(x[0, c*n:(c+1)*n].copy(), x[1, c*n:(c+1)*n].copy()))
        in_len += n
        k = in_len // 512
        in_len -= 512 * k
        out_len += 512 * k
        if out_len >= n:
            out_len -= n
            served.append(g)
        else:
            silent += n
            assert np.all(g == 0), c
    assert eng.silent_frames() == silent == eng.latency_samples(n) and silent > 0
    got = np.concatenate(served, 1)
    assert_parity(got, x[:, :got.shape[1]], "delta HRIR: the served calls are the input stream, end to end")


@pytest.mark.parametrize("nb_on", [1, 4, 10, 12])
def test_chain_with_eq_through_the_resident_kernel_is_bit_identical_and_matches_the_oracle(ohs, synth, oracle, nb_on):
    """lib.rs:1179-1207 served by the resident kernel: EQ (ring form, <= 12 enabled bands) -> convolution -> gain in
    one mailbox round trip; per-call coefficient refresh as the plugin does it (the EQ block is re-sent when it
    changes); same bits as the launch-per-call chain, parity with the oracle."""
    bands = synth.eq_table()
    irs = synth.hrir_set(512)

    def make(rt):
        e = ohs.ConvolutionEngine.new()
        for p in range(4):
            e.set_ir(p, irs[p])
        q = ohs.StereoParametricEQ.new(len(bands) + 2, synth.FS)
        for i, b in enumerate(bands[:nb_on]):
            q.update_band_coeffs(i, synth.FS, b)
        if rt:
            e.set_realtime(True)
        return e, q

    er, qr = make(True)
    el, ql = make(False)
    eo = oracle.ConvolutionEngine()
    for p in range(4):
        eo.set_ir(p, irs[p])
    qo = oracle.StereoParametricEQ(len(bands) + 2, synth.FS)
    for i in range(nb_on):
        c, en = qr.get_band_coeffs(i)
        qo.set_band_coeffs(i, c, en)
    sizes = [1024, 1024, 512, 256, 100, 2048, 1024, 37, 1024, 4096, 1024]
    x = synth.white_noise([77], sum(sizes))[0]
    o = 0
    got, want = [], []
    for k, n in enumerate(sizes):
        if k == 4:                                  # a parameter change between callbacks: band 0 retuned, band 1 off / on
            nb = ohs.BandConfig(ohs.FilterType.Peak, 900.0, 1.1, -5.0, True)
            for q in (qr, ql):
                q.update_band_coeffs(0, synth.FS, nb)
            qo.set_band_coeffs(0, qr.get_band_coeffs(0)[0], True)
        gain = 0.5 + 0.05 * k
        a_l, a_r = x[0, o:o + n].copy(), x[1, o:o + n].copy()
        b_l, b_r = a_l.copy(), a_r.copy()
        c_l, c_r = a_l.copy(), a_r.copy()
        ohs.process_chain(er, qr, a_l, a_r, eq_enable=True, output_gain=gain)
        ohs.process_chain(el, ql, b_l, b_r, eq_enable=True, output_gain=gain)
        oracle.chain_process(eo, qo, c_l, c_r, eq_enable=True, gain=gain)
        assert np.array_equal(a_l.view(np.uint32), b_l.view(np.uint32)) and np.array_equal(a_r.view(np.uint32), b_r.view(np.uint32)), (k, n)
        got.append(np.stack([a_l, a_r])); want.append(np.stack([c_l, c_r]))
        o += n
    assert_parity(np.concatenate(got, 1), np.concatenate(want, 1), "resident chain vs oracle")
    # EQ disabled / bypass through the same entry point
    a_l, a_r = x[0, :1024].copy(), x[1, :1024].copy()
    b_l, b_r = a_l.copy(), a_r.copy()
    ohs.process_chain(er, qr, a_l, a_r, eq_enable=False, output_gain=1.0)
    ohs.process_chain(el, ql, b_l, b_r, eq_enable=False, output_gain=1.0)
    assert np.array_equal(a_l, b_l) and np.array_equal(a_r, b_r)
    keep = a_l.copy()
    ohs.process_chain(er, qr, a_l, a_r, master_bypass=True, eq_enable=True)
    assert np.array_equal(a_l, keep)


def test_device_wide_waits_of_other_handles_do_not_stall_on_resident_kernels(ohs, synth):
    """hipFree / hipDeviceSynchronize inside the library wait for every kernel on the device -- a resident kernel of
    ANOTHER engine included, which would only leave after its ~2 s idle limit.  Library code about to make such a wait
    asks the device's resident kernels to leave first (and holds new ones off meanwhile); their owners restart them
    on the next call.  Two engines in real-time mode + set_ir / destroy / a batch's set_ir and reset on other handles:
    nothing may take anywhere near the idle limit, and the real-time engines keep producing the right bits."""
    a, ref_a, irs = _pair(ohs, synth, 512)
    b, ref_b, _ = _pair(ohs, synth, 512)
    a.set_realtime(True); b.set_realtime(True)
    x = synth.white_noise([21], 64 * 1024)[0]
    pos = 0

    def step(n=1024):
        nonlocal pos
        l, r = x[0, pos:pos + n].copy(), x[1, pos:pos + n].copy()
        for rt, ref in ((a, ref_a), (b, ref_b)):
            g, e = rt.process_block(l, r), ref.process_block(l, r)
            assert np.array_equal(np.stack(g).view(np.uint32), np.stack(e).view(np.uint32))
        pos += n

    for _ in range(4):
        step()                                  # both resident kernels are up
    c = ohs.ConvolutionEngine.new()
    t0 = time.perf_counter()
    for p in range(4):
        c.set_ir(p, irs[p][:300])               # hipFree + stream syncs on a third handle
    c.process_block(x[0, :20000].copy(), x[1, :20000].copy())     # grows FIFOs / staging buffers (more frees)
    bp = ohs.BatchProcessor(3, num_bands=2)
    for p in range(4):
        bp.set_ir(p, irs[p])                    # hipDeviceSynchronize
    bp.reset()
    del c, bp                                   # destroy paths
    import gc
    gc.collect()
    dt = time.perf_counter() - t0
    assert dt < 1.0, f"device-wide waits took {dt:.2f} s: they sat behind a resident kernel"
    for _ in range(6):
        step()                                  # restarted on their next calls, state intact
    # ... and while one engine keeps its doorbell ringing from another thread
    import threading
    stop = threading.Event()
    errs = []

    def ring():
        rt2, ref2, _ = _pair(ohs, synth, 256)
        rt2.set_realtime(True)
        y = synth.white_noise([22], 1024)[0]
        try:
            while not stop.is_set():
                g, e = rt2.process_block(y[0], y[1]), ref2.process_block(y[0], y[1])
                if not np.array_equal(np.stack(g).view(np.uint32), np.stack(e).view(np.uint32)):
                    errs.append("mismatch")
                    return
        except Exception as ex:     # noqa: BLE001
            errs.append(repr(ex))

    th = threading.Thread(target=ring)
    th.start()
    time.sleep(0.2)
    t0 = time.perf_counter()
    for k in range(5):
        d = ohs.ConvolutionEngine.new()
        d.set_ir(0, irs[0][:100 + k])
        del d
    gc.collect()
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    assert not errs, errs
    assert dt < 2.0, f"create / set_ir / destroy next to a busy real-time engine took {dt:.2f} s"
