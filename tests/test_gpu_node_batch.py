"""The batch mode over the GPUs of a node behind the C ABI (ohs_node_batch_*): on this one-GPU box the RCCL
communicator has size 1 -- ncclCommInitAll, the grouped ncclBroadcast of the packed tables and the unpacking on the
device all run -- and the results must be bit-identical to the single-device batch (ohs_batch_process_host) fed the
same tables: streams do not interact, so sharding changes nothing."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ohs():
    import open_headstage_amd
    return open_headstage_amd


@pytest.fixture(scope="module")
def synth():
    from open_headstage_amd import synth as s
    return s


def _tables(ohs, synth, taps):
    irs = synth.hrir_set(taps)
    bands = synth.eq_table()
    coeffs = np.stack([ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db) for b in bands])
    enabled = np.array([b.enabled for b in bands])
    return irs, coeffs, enabled


@pytest.mark.parametrize("taps", [512, 1300])
def test_node_batch_one_device_bit_identical_to_the_device_batch(ohs, synth, oracle, taps):
    S, n_blocks = 7, 96
    irs, coeffs, enabled = _tables(ohs, synth, taps)
    nb = ohs.NodeBatchProcessor(S, num_bands=len(coeffs), n_devices=1)
    assert nb.shard(0) == (0, 0, S)
    assert nb.rccl_info() == (1, True)              # a real communicator, from the loaded librccl.so
    nb.set_tables(irs, coeffs, enabled)
    nb.set_eq_enabled(True)
    nb.set_gain(0.7)
    x = synth.white_noise(range(S), n_blocks * 512)
    y_node = nb.process_host(x.copy(), chunk_blocks=32)
    bp = ohs.BatchProcessor(S, num_bands=len(coeffs))
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i in range(len(coeffs)):
        bp.set_band_coeffs(i, coeffs[i], bool(enabled[i]))
    bp.set_eq_enabled(True)
    bp.set_gain(0.7)
    y_one = bp.process_host(x.copy(), chunk_blocks=32)
    assert np.array_equal(y_node.view(np.uint32), y_one.view(np.uint32))
    # and within the parity bar of the oracle (stream 3)
    eng = oracle.ConvolutionEngine()
    for p in range(4):
        eng.set_ir(p, irs[p])
    eq = oracle.StereoParametricEQ(len(coeffs), synth.FS)
    for i in range(len(coeffs)):
        eq.set_band_coeffs(i, coeffs[i], bool(enabled[i]))
    l, r = x[3, 0].copy(), x[3, 1].copy()
    oracle.chain_process(eng, eq, l, r, eq_enable=True, gain=0.7)
    ref = np.stack([l, r]).astype(np.float64)
    err = np.sqrt(np.mean((y_node[3] - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))
    assert err <= 1e-6, err
    # a second call continues the streams; one path changed in mid-stream through the broadcast path
    nb.set_ir(2, irs[2][:200])
    bp.set_ir(2, irs[2][:200])
    x2 = synth.white_noise(range(100, 100 + S), n_blocks * 512)
    assert np.array_equal(nb.process_host(x2.copy(), chunk_blocks=32).view(np.uint32),
                          bp.process_host(x2.copy(), chunk_blocks=32).view(np.uint32))
    # reset returns both to the same fresh state (tables kept: path 2 stays the shortened response)
    nb.reset(); bp.reset()
    y3 = nb.process_host(x.copy(), chunk_blocks=32)
    assert np.array_equal(y3.view(np.uint32), bp.process_host(x.copy(), chunk_blocks=32).view(np.uint32))
    assert not np.array_equal(y3, y_one)


def test_node_batch_device_resident_process_is_the_device_batch_bit_for_bit(ohs, synth):
    """ohs_node_batch_process: per-device device pointers, each device's launches issued by its own host thread on its
    own stream, ohs_node_batch_sync, the per-slot device timer -- at one device the bits of ohs_batch_process."""
    import torch
    from open_headstage_amd.batch import device_pci_bus_id
    dev = torch.device("cuda:0")
    S, n_blocks = 19, 130
    irs, coeffs, enabled = _tables(ohs, synth, 512)
    nb = ohs.NodeBatchProcessor(S, num_bands=len(coeffs), n_devices=1)
    nb.set_tables(irs, coeffs, enabled)
    nb.set_eq_enabled(True)
    nb.set_gain(0.8)
    bp = ohs.BatchProcessor(S, num_bands=len(coeffs))
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i in range(len(coeffs)):
        bp.set_band_coeffs(i, coeffs[i], bool(enabled[i]))
    bp.set_eq_enabled(True)
    bp.set_gain(0.8)
    assert nb.stream(0) != 0
    bus = device_pci_bus_id(0)
    assert bus.count(":") == 2, bus
    for call in range(3):
        x = synth.white_noise_torch(50 * call, S, n_blocks * 512, dev)
        y_node, y_one = torch.empty_like(x), torch.empty_like(x)
        torch.cuda.synchronize()
        nb.timer_begin()
        nb.process([x], [y_node])
        nb.process([y_node], [y_node])          # calls queue up in order; in place allowed
        ms = nb.timer_end()
        assert len(ms) == 1 and ms[0] > 0.0
        nb.sync()
        bp.process(x, out=y_one)
        bp.process(y_one, out=y_one)
        torch.cuda.synchronize()
        assert torch.equal(y_node.view(torch.int32), y_one.view(torch.int32)), call
        if call == 0:       # tables changed between calls: the set_* calls wait for what is queued
            nb.set_ir(1, irs[1][:300]); bp.set_ir(1, irs[1][:300])
    with pytest.raises(ohs.OhsError):
        nb.process_ptrs([0], [0], 4, 4096, 2048)        # NULL device pointers


@pytest.mark.parametrize("slots,S", [(2, 9), (3, 10), (4, 4)])
def test_node_batch_several_slots_rehearsed_on_one_device(ohs, synth, exp_tuning, slots, S):
    """What a one-GPU box can check of N > 1: the experiments library lets several device slots name the SAME device
    (Tuning::node_shared_device_rehearsal), so the sharding, the per-slot unpack of the packed tables, the per-slot worker
    threads, streams and pointers and the shard offsets of process_host all run with N slots -- against one batch of all
    the streams, bit for bit.  NOT covered: the grouped ncclBroadcast over N communicators (RCCL refuses two ranks on a
    device; in this mode the tables travel by device copies and rccl_info reports size 0)."""
    import torch
    dev = torch.device("cuda:0")
    L = exp_tuning.lib
    exp_tuning("node_shared_device_rehearsal", "1")
    irs, coeffs, enabled = _tables(ohs, synth, 1300)
    nb = ohs.NodeBatchProcessor(S, num_bands=len(coeffs), devices=[0] * slots, library=L)
    assert nb.rccl_info()[0] == 0
    shards = [nb.shard(i) for i in range(slots)]
    assert [s[0] for s in shards] == [0] * slots
    assert sum(s[2] for s in shards) == S and [s[1] for s in shards] == list(np.cumsum([0] + [s[2] for s in shards[:-1]]))
    assert max(s[2] for s in shards) - min(s[2] for s in shards) <= 1
    nb.set_tables(irs, coeffs, enabled)
    nb.set_eq_enabled(True)
    nb.set_gain(0.9)
    bp = ohs.BatchProcessor(S, num_bands=len(coeffs), library=L)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i in range(len(coeffs)):
        bp.set_band_coeffs(i, coeffs[i], bool(enabled[i]))
    bp.set_eq_enabled(True)
    bp.set_gain(0.9)
    n = 40 * 512
    for call in range(3):
        x = synth.white_noise_torch(9 * call, S, n, dev)
        xs = [x[f:f + c].contiguous() for _, f, c in shards]
        ys = [torch.empty_like(a) for a in xs]
        torch.cuda.synchronize()
        nb.timer_begin()
        nb.process(xs, ys)
        ms = nb.timer_end()
        assert len(ms) == slots and all(m > 0.0 for m in ms)
        nb.sync()
        y_one = bp.process(x)
        torch.cuda.synchronize()
        assert torch.equal(torch.cat(ys).view(torch.int32), y_one.view(torch.int32)), call
        if call == 0:       # one path and one band changed in mid-stream, on every slot
            nb.set_ir(3, irs[3][:700]); bp.set_ir(3, irs[3][:700])
        if call == 1:
            nb.set_conv_plan(2); bp.set_conv_plan(2)
    # the host-buffer form: every slot takes its own rows of the one host array
    xh = synth.white_noise(range(300, 300 + S), n)
    assert np.array_equal(nb.process_host(xh.copy(), chunk_blocks=16).view(np.uint32),
                          bp.process_host(xh.copy(), chunk_blocks=16).view(np.uint32))
    nb.reset(); bp.reset()
    assert np.array_equal(nb.process_host(xh.copy(), chunk_blocks=16).view(np.uint32),
                          bp.process_host(xh.copy(), chunk_blocks=16).view(np.uint32))


def _rehearsal_pair(ohs, synth, exp_tuning, slots, S, taps=512):
    L = exp_tuning.lib
    exp_tuning("node_shared_device_rehearsal", "1")
    irs, coeffs, enabled = _tables(ohs, synth, taps)
    nb = ohs.NodeBatchProcessor(S, num_bands=len(coeffs), devices=[0] * slots, library=L)
    nb.set_tables(irs, coeffs, enabled)
    nb.set_eq_enabled(True)
    bp = ohs.BatchProcessor(S, num_bands=len(coeffs), library=L)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i in range(len(coeffs)):
        bp.set_band_coeffs(i, coeffs[i], bool(enabled[i]))
    bp.set_eq_enabled(True)
    return nb, bp, (irs, coeffs, enabled)


def test_default_plan_follows_the_streams_per_device_so_shards_agree_to_rounding_not_bit_for_bit(ohs, synth, exp_tuning):
    """ADVICE r04 / the header note of ohs_node_batch_process_host: with the library's own plan choice (plan 0) an UNSHARDED
    batch of 512 streams takes the hop-1536 plan for calls of 48 blocks or more (out of place, EQ off: from 256 streams),
    its shards of 256 streams or fewer per device may not -- here 3 slots of 171 / 171 / 170 streams keep the block-512
    plan.  The two plans agree to f32 rounding (both 2-3e-7 from the truth), not bit for bit; with the plan PINNED on both
    sides the shards and the unsharded batch agree bit for bit again.  This is the documented behaviour, held here."""
    import torch
    dev = torch.device("cuda:0")
    slots, S, n = 3, 512, 48 * 512
    L = exp_tuning.lib
    exp_tuning("node_shared_device_rehearsal", "1")
    irs = synth.hrir_set(512)
    x = synth.white_noise_torch(33, 4, n, dev)[torch.arange(S, device=dev) % 4].contiguous()
    out = {}
    for plan in (0, 1):
        nb = ohs.NodeBatchProcessor(S, num_bands=10, devices=[0] * slots, library=L)
        nb.set_tables(irs)
        nb.set_eq_enabled(False)
        nb.set_conv_plan(plan)
        bp = ohs.BatchProcessor(S, num_bands=10, library=L)
        for p in range(4):
            bp.set_ir(p, irs[p])
        bp.set_eq_enabled(False)
        bp.set_conv_plan(plan)
        shards = [nb.shard(i) for i in range(slots)]
        xs = [x[f:f + c].contiguous() for _, f, c in shards]
        ys = [torch.empty_like(a) for a in xs]
        torch.cuda.synchronize()
        nb.process(xs, ys); nb.sync()
        y_one = torch.empty_like(x)
        bp.process(x, out=y_one)
        torch.cuda.synchronize()
        out[plan] = (torch.cat(ys), y_one, bp.last_conv_plan()[0], nb.device_batch(0).last_conv_plan()[0])
    ys0, one0, fam_one0, fam_shard0 = out[0]
    assert fam_one0 == "hop1536_p1" and fam_shard0 == "block512_p1", (fam_one0, fam_shard0)
    assert not torch.equal(ys0.view(torch.int32), one0.view(torch.int32))       # different plans: different rounding ...
    d = (ys0 - one0).double()
    rel = float(torch.sqrt((d * d).mean()) / torch.sqrt((one0.double() ** 2).mean()))
    assert rel <= 1e-6, rel                                                     # ... inside the parity bar
    ys1, one1, fam_one1, fam_shard1 = out[1]
    assert fam_one1 == fam_shard1 == "block512_p1"
    assert torch.equal(ys1.view(torch.int32), one1.view(torch.int32))           # pinned: bit for bit
    assert torch.equal(ys1.view(torch.int32), ys0.view(torch.int32))            # (the shards ran the block-512 plan both times)


def test_node_batch_fails_closed_when_one_slot_fails(ohs, synth, exp_tuning):
    """A processing call that fails on ONE device slot leaves the other shards advanced: the node handle refuses every
    later processing call (device-resident and host-buffer form) until ohs_node_batch_reset, then answers with the bits
    of a fresh object.  The failure is injected in the experiments build (the k-th chunk launch of any batch fails)."""
    import torch
    from open_headstage_amd import _ffi
    dev = torch.device("cuda:0")
    slots, S, n = 3, 9, 128 * 512
    nb, bp, _ = _rehearsal_pair(ohs, synth, exp_tuning, slots, S)
    shards = [nb.shard(i) for i in range(slots)]
    x = synth.white_noise_torch(4, S, n, dev)
    xs = [x[f:f + c].contiguous() for _, f, c in shards]
    ys = [torch.empty_like(a) for a in xs]
    y_fresh = bp.process(x).clone()
    torch.cuda.synchronize()
    exp_tuning.lib.ohs_debug_inject_batch_failure(8)       # 6 time chunks per slot and call: some slot's second chunk or later
    with pytest.raises(ohs.OhsError) as ei:
        nb.process(xs, ys)
        nb.sync()
    assert ei.value.status == _ffi.OHS_ERR_HIP and "device slot" in str(ei.value) and "injected failure" in str(ei.value)
    exp_tuning.lib.ohs_debug_inject_batch_failure(0)
    torch.cuda.synchronize()
    for call in (lambda: nb.process(xs, ys), lambda: nb.process_host(x.cpu().numpy())):
        with pytest.raises(ohs.OhsError) as ei:
            call()
        assert ei.value.status == _ffi.OHS_ERR_HIP and "out of step" in str(ei.value) and "ohs_node_batch_reset" in str(ei.value)
    # sync / the timer do not answer "OK" over half-advanced state either (they still wait for every device)
    for call in (nb.sync, nb.timer_begin, nb.timer_end):
        with pytest.raises(ohs.OhsError) as ei:
            call()
        assert ei.value.status == _ffi.OHS_ERR_HIP and "out of step" in str(ei.value)
    nb.reset()
    nb.process(xs, ys)
    nb.sync()
    assert torch.equal(torch.cat(ys).view(torch.int32), y_fresh.view(torch.int32))


def test_node_batch_argument_errors_do_not_poison_the_handle(ohs, synth, exp_tuning):
    """an OHS_ERR_INVALID_ARG is found by every slot before anything is queued or stored -- the node batch
    must stay usable (no reset, which would zero every stream's state; no repeated table call) and answer the next
    valid call with the bits of the unsharded batch."""
    import ctypes as C
    import torch
    from open_headstage_amd import _ffi
    dev = torch.device("cuda:0")
    slots, S, n = 3, 7, 16 * 512
    nb, bp, _ = _rehearsal_pair(ohs, synth, exp_tuning, slots, S)
    shards = [nb.shard(i) for i in range(slots)]
    x = synth.white_noise_torch(5, S, 2 * n, dev)
    xa, xb = x[:, :, :n].contiguous(), x[:, :, n:].contiguous()
    torch.cuda.synchronize()
    ys = [torch.empty_like(xa[f:f + c]) for _, f, c in shards]
    nb.process([xa[f:f + c].contiguous() for _, f, c in shards], ys); nb.sync()
    # strides smaller than the processed region, then a block count beyond 2^24: refused by every slot
    xs = [xb[f:f + c].contiguous() for _, f, c in shards]
    with pytest.raises(ohs.OhsError) as ei:
        nb.process_ptrs([t.data_ptr() for t in xs], [t.data_ptr() for t in ys], n // 512, 2 * n, n // 2)
    assert ei.value.status == _ffi.OHS_ERR_INVALID_ARG
    with pytest.raises(ohs.OhsError) as ei:
        nb.process_ptrs([t.data_ptr() for t in xs], [t.data_ptr() for t in ys], (1 << 24) + 1, 1 << 40, 1 << 39)
    assert ei.value.status == _ffi.OHS_ERR_INVALID_ARG
    # a NULL coefficient pointer: refused, the tables are not torn
    L = nb._lib
    assert L.ohs_node_batch_set_eq_band_coeffs(nb._h, C.c_size_t(0), None, 1) == _ffi.OHS_ERR_INVALID_ARG
    nb.process(xs, ys); nb.sync()          # continues where the first call stopped: no reset happened
    ya = bp.process(xa); yb = bp.process(xb)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(ys).view(torch.int32), yb.view(torch.int32))


def test_node_batch_refuses_to_process_on_torn_tables(ohs, synth, exp_tuning):
    """A table call that fails after some devices have taken the new tables leaves the node with different tables per
    device: processing is refused until a complete ohs_node_batch_set_tables has succeeded."""
    import torch
    from open_headstage_amd import _ffi
    dev = torch.device("cuda:0")
    slots, S, n = 3, 7, 24 * 512
    nb, bp, (irs, coeffs, enabled) = _rehearsal_pair(ohs, synth, exp_tuning, slots, S)
    shards = [nb.shard(i) for i in range(slots)]
    x = synth.white_noise_torch(11, S, n, dev)
    xs = [x[f:f + c].contiguous() for _, f, c in shards]
    ys = [torch.empty_like(a) for a in xs]
    torch.cuda.synchronize()
    nb.process(xs, ys); nb.sync()
    y1 = bp.process(x)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(ys).view(torch.int32), y1.view(torch.int32))
    irs2 = [h[::-1].copy() * 0.5 for h in irs]
    exp_tuning("node_inject_table_failure_slot", "2")       # slot 0 takes the new tables, slot 1 fails, slot 2 keeps the old
    with pytest.raises(ohs.OhsError) as ei:
        nb.set_tables(irs2, coeffs, enabled)
    assert "injected table failure" in str(ei.value)
    exp_tuning("node_inject_table_failure_slot", "0")
    with pytest.raises(ohs.OhsError) as ei:
        nb.process(xs, ys)
    assert ei.value.status == _ffi.OHS_ERR_HIP and "different" in str(ei.value) and "ohs_node_batch_set_tables" in str(ei.value)
    nb.set_ir(0, irs2[0])                                   # one path alone does not repair it
    with pytest.raises(ohs.OhsError):
        nb.process(xs, ys)
    nb.set_tables(irs2, coeffs, enabled)                    # the complete set does; every path restarts (convolution.rs:135-138)
    for p in range(4):
        bp.set_ir(p, irs2[p])
    nb.process(xs, ys); nb.sync()
    y2 = bp.process(x)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(ys).view(torch.int32), y2.view(torch.int32))


def test_node_batch_without_librccl_fails_with_err_rccl(tmp_path):
    """OHS_ERR_RCCL when librccl.so cannot be loaded: a child process whose dynamic loader is made to miss it.  The
    library names it tries are librccl.so.1 / librccl.so by soname and under /opt/rocm/lib; the child pre-loads a
    dlopen() that refuses every name containing "rccl"."""
    import os
    import subprocess
    import sys
    src = tmp_path / "nodlopen.c"
    src.write_text(
        '#define _GNU_SOURCE\n#include <dlfcn.h>\n#include <string.h>\n#include <stddef.h>\n'
        'void *dlopen(const char *name, int flags)\n{\n'
        '    static void *(*real)(const char *, int);\n'
        '    if (!real) real = (void *(*)(const char *, int))dlsym(RTLD_NEXT, "dlopen");\n'
        '    if (name && strstr(name, "rccl")) return NULL;\n'
        '    return real(name, flags);\n}\n')
    so = tmp_path / "libnodlopen.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", str(so), str(src), "-ldl"], check=True)
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import open_headstage_amd as ohs\n"
        "from open_headstage_amd import _ffi\n"
        "try:\n"
        "    ohs.NodeBatchProcessor(4, n_devices=1)\n"
        "    print('CREATED')\n"
        "except ohs.OhsError as e:\n"
        "    print('STATUS', e.status, str(e))\n"
        "bp = ohs.BatchProcessor(2)\n"          # the single-device batch does not need RCCL
        "print('BATCH_OK')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, LD_PRELOAD=str(so))
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "STATUS 9" in res.stdout and "librccl" in res.stdout, res.stdout     # OHS_ERR_RCCL
    assert "BATCH_OK" in res.stdout


def test_node_batch_argument_checks(ohs):
    from open_headstage_amd import _ffi
    with pytest.raises(ohs.OhsError) as ei:
        ohs.NodeBatchProcessor(1, n_devices=2)          # fewer streams than devices
    assert ei.value.status == _ffi.OHS_ERR_INVALID_ARG
    with pytest.raises(ohs.OhsError):
        ohs.NodeBatchProcessor(8, devices=[0, 0])       # a device listed twice
    with pytest.raises(ohs.OhsError):
        ohs.NodeBatchProcessor(8, devices=[99])         # no such device
