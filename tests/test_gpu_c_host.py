"""A compiled host on the C ABI: examples/host_chain.c (the C twin of the Rust shim in INTEGRATION.md)
is built with gcc against include/ohs_hip.h + libohs_hip.so, run as its own process on the GPU, and
its output is checked against the oracle -- no Python, ctypes or torch between the host and the library."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.util import assert_parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_chain_matches_oracle(tmp_path, oracle):
    import open_headstage_amd as ohs
    from open_headstage_amd import synth

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this box")
    libdir = os.path.join(ROOT, "open_headstage_amd")
    exe = str(tmp_path / "host_chain")
    subprocess.run([gcc, "-O2", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "examples", "host_chain.c"), "-L" + libdir, "-lohs_hip", "-lm",
                    "-Wl,-rpath," + libdir], check=True)
    taps, frames = 512, 1024 * 12 + 300          # a ragged last host block
    irs = synth.hrir_set(taps)
    x = synth.white_noise([77], frames)[0]
    np.stack(irs).astype(np.float32).tofile(tmp_path / "irs.f32")
    x.astype(np.float32).tofile(tmp_path / "in.f32")
    r = subprocess.run([exe, str(tmp_path / "irs.f32"), str(taps), str(tmp_path / "in.f32"),
                        str(tmp_path / "out.f32"), str(frames)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    y = np.fromfile(tmp_path / "out.f32", np.float32).reshape(2, frames)

    eng = oracle.ConvolutionEngine()
    for p in range(4):
        eng.set_ir(p, irs[p])
    bands = synth.eq_table()
    eq = oracle.StereoParametricEQ(len(bands), synth.FS)
    for i, b in enumerate(bands):
        # the library's own coefficient bits (coefficients are inputs to the EQ)
        eq.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), True)
    ref = np.empty_like(y)
    for pos in range(0, frames, 1024):
        l, rr = x[0, pos:pos + 1024].copy(), x[1, pos:pos + 1024].copy()
        oracle.chain_process(eng, eq, l, rr, eq_enable=True, gain=0.8912509)
        ref[0, pos:pos + 1024], ref[1, pos:pos + 1024] = l, rr
    assert_parity(y, ref, "C host chain")


def test_c_host_node_batch_matches_oracle(tmp_path, oracle):
    """examples/host_node_batch.c: the node-level batch API (RCCL communicator of the box's one GPU) from a compiled C
    host, against the oracle."""
    import open_headstage_amd as ohs
    from open_headstage_amd import synth

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this box")
    libdir = os.path.join(ROOT, "open_headstage_amd")
    exe = str(tmp_path / "host_node_batch")
    subprocess.run([gcc, "-O2", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "examples", "host_node_batch.c"), "-L" + libdir, "-lohs_hip", "-lm",
                    "-Wl,-rpath," + libdir], check=True)
    taps, frames, S = 512, 512 * 40, 5
    irs = synth.hrir_set(taps)
    x = synth.white_noise(range(300, 300 + S), frames)
    np.stack(irs).astype(np.float32).tofile(tmp_path / "irs.f32")
    x.astype(np.float32).tofile(tmp_path / "in.f32")
    r = subprocess.run([exe, "1", str(S), str(tmp_path / "irs.f32"), str(taps), str(tmp_path / "in.f32"),
                        str(tmp_path / "out.f32"), str(frames)], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RCCL communicator of 1 rank(s)" in r.stderr and "streams [0, 5)" in r.stderr
    y = np.fromfile(tmp_path / "out.f32", np.float32).reshape(S, 2, frames)
    bands = synth.eq_table()
    for s in range(S):
        eng = oracle.ConvolutionEngine()
        for p in range(4):
            eng.set_ir(p, irs[p])
        eq = oracle.StereoParametricEQ(len(bands), synth.FS)
        for i, b in enumerate(bands):
            eq.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), True)
        l, rr = x[s, 0].copy(), x[s, 1].copy()
        oracle.chain_process(eng, eq, l, rr, eq_enable=True, gain=0.8912509)
        assert_parity(y[s], np.stack([l, rr]), f"C node-batch host, stream {s}")
    # the same host built -DDEVICE_RESIDENT: it owns the device buffers (hipMalloc from C) and drives the node's data
    # path, ohs_node_batch_process / _timer_end -- the bits of the PCIe-fed form (streams do not interact, the time
    # chunking of the host pipeline changes nothing)
    rocm = "/opt/rocm"
    if not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("no HIP headers for the device-resident C host")
    exe2 = str(tmp_path / "host_node_batch_dev")
    subprocess.run([gcc, "-O2", "-std=c99", "-Wall", "-DDEVICE_RESIDENT", "-D__HIP_PLATFORM_AMD__",
                    "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(rocm, "include"), "-o", exe2,
                    os.path.join(ROOT, "examples", "host_node_batch.c"), "-L" + libdir, "-lohs_hip",
                    "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lm",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(rocm, "lib")], check=True)
    r = subprocess.run([exe2, "1", str(S), str(tmp_path / "irs.f32"), str(taps), str(tmp_path / "in.f32"),
                        str(tmp_path / "out_dev.f32"), str(frames)], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ms on the device" in r.stderr
    y_dev = np.fromfile(tmp_path / "out_dev.f32", np.float32).reshape(S, 2, frames)
    assert np.array_equal(y_dev.view(np.uint32), y.view(np.uint32))
