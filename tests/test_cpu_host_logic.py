"""CPU suite (no GPU): the C-ABI library loads and exports what include/ohs_hip.h declares, fails
loudly without a device, and the host-side helpers (synthetic generators, stream sharding, table
packing, world_size-2 gloo broadcast + sharded run) behave."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from open_headstage_amd import build
    build.build()
    from open_headstage_amd import _ffi
    return _ffi.lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "ohs_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(ohs_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    from open_headstage_amd import _ffi
    assert sorted(_ffi.PROTOTYPES) == declared, "ctypes prototypes and header disagree"
    for name in declared:
        assert hasattr(lib, name), name
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "open_headstage_amd", "libohs_hip.so")],
                        capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (ohs_[a-z0-9_]+)", nm))
    assert set(declared) <= exported
    # ... and NOTHING else: every defined dynamic symbol of the product library is an entry of the header;
    # the helpers its translation units share (ohsint_*) and the C++ internals are local (csrc/exports.map)
    defined = {l.split()[-1] for l in nm.splitlines() if len(l.split()) >= 3 and l.split()[-2] in "TDBRWV"}
    assert defined == set(declared), sorted(defined ^ set(declared))


def test_no_cpu_fallback_without_device(lib):
    """On a box without a GPU every create call must fail with OHS_ERR_NO_DEVICE (never compute)."""
    import ctypes as C
    from open_headstage_amd import _ffi
    n = C.c_int()
    rc = lib.ohs_device_count(C.byref(n))
    if rc == _ffi.OHS_OK:
        pytest.skip("a GPU is visible here")
    assert rc == _ffi.OHS_ERR_NO_DEVICE
    import open_headstage_amd as ohs
    for ctor in (lambda: ohs.ConvolutionEngine.new(), lambda: ohs.StereoParametricEQ.new(10, 48000.0),
                 lambda: ohs.BatchProcessor(4), lambda: ohs.NodeBatchProcessor(16, n_devices=2)):
        with pytest.raises(ohs.OhsError) as ei:
            ctor()
        assert ei.value.status == _ffi.OHS_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.ohs_last_error()


def test_product_and_oracle_coefficient_routines_are_one_build(lib, oracle):
    """A BUILD-CONSISTENCY check, not parity evidence: the product's and the oracle's coefficient routines are the same
    formulas on byte-identical copies of libm_musl.h, so this compares a text with itself -- it only catches the two
    copies or their compile flags drifting apart.  What the values ARE is pinned independently in
    tests/test_cpu_coeff_pins.py (correctly rounded libm results, a numpy re-implementation, the f64 cookbook)."""
    from open_headstage_amd import FilterType, OhsError, biquad_coefficients
    for t in range(8):
        for fc, q, g in [(20.0, 0.707, 0.0), (105.0, 0.7, 4.0), (1000.0, 1.41, -3.0), (23999.0, 10.0, 16.0)]:
            a = biquad_coefficients(FilterType(t), 48000.0, fc, q, g)
            b = oracle.biquad_coeffs(t, 48000.0, fc, q, g)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (t, fc, q, g)
    with pytest.raises(OhsError):
        biquad_coefficients(FilterType.Peak, 48000.0, 24001.0, 1.0, 0.0)   # OutsideNyquist
    with pytest.raises(OhsError):
        biquad_coefficients(FilterType.Peak, 48000.0, 1000.0, -1.0, 0.0)    # NegativeQ


def test_synth_generators_are_deterministic():
    from open_headstage_amd import synth
    assert int(synth.splitmix64(0, 1)[0]) == 0xE220A8397B1DCDAF       # published splitmix64 vector
    assert int(synth.splitmix64(0, 3)[2]) == int(synth.splitmix64(0, 1, offset=2)[0])
    x = synth.white_noise([0, 1], 4096)
    assert x.dtype == np.float32 and x.shape == (2, 2, 4096)
    assert x.min() >= -1.0 and x.max() < 1.0 and abs(float(x.mean())) < 0.02
    assert not np.array_equal(x[0, 0], x[0, 1]) and not np.array_equal(x[0, 0], x[1, 0])
    y = synth.white_noise([1], 100, offset=50)
    assert np.array_equal(y[0], x[1, :, 50:150])
    for taps in (128, 512, 16384):
        h = synth.hrir_set(taps)
        assert all(a.size == taps and a.dtype == np.float32 for a in h)
        assert abs(np.abs(h[0]).sum() + np.abs(h[2]).sum() - 1) < 1e-5      # |y| <= 1 for |x| <= 1
        assert abs(np.abs(h[1]).sum() + np.abs(h[3]).sum() - 1) < 1e-5
    bands = synth.eq_table()
    assert len(bands) == 10 and all(b.enabled for b in bands)


def test_shard_range_partitions_streams():
    from open_headstage_amd.sharding import shard_range
    for n, w in [(2048, 8), (2048, 1), (10, 4), (3, 8), (0, 2)]:
        got = [i for r in range(w) for i in shard_range(n, r, w)]
        assert got == list(range(n))
        sizes = [len(shard_range(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_table_pack_roundtrip():
    from open_headstage_amd import sharding, synth
    irs = synth.hrir_set(128)
    c = np.arange(50, dtype=np.float32).reshape(10, 5)
    en = np.array([1, 0, 1, 1, 1, 0, 1, 1, 1, 1], bool)
    buf = sharding.pack_tables(irs, c, en)
    assert buf.size == sharding.table_numel(128, 10)
    irs2, c2, en2 = sharding.unpack_tables(buf)
    assert all(np.array_equal(a, b) for a, b in zip(irs, irs2))
    assert np.array_equal(c, c2) and np.array_equal(en, en2)


_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from open_headstage_amd import sharding, synth
from oracle import ohs_oracle as orc     # tests may use the oracle as the stand-in executor

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
TAPS, NB, S, FRAMES = 128, 10, 6, 2048
if rank == 0:
    irs = synth.hrir_set(TAPS)
    bands = synth.eq_table()
    coeffs = np.stack([orc.biquad_coeffs(int(b.filter_type), synth.FS, b.center_freq, b.q, b.gain_db) for b in bands])
    en = np.array([b.enabled for b in bands])
else:
    irs = coeffs = en = None
irs, coeffs, en = sharding.broadcast_tables(irs, coeffs, en, TAPS, NB, torch.device("cpu"), src=0)
ids = sharding.shard_range(S, rank, world)
x = synth.white_noise(ids, FRAMES)
out = np.zeros((S, 2, FRAMES), np.float32)
for k, sid in enumerate(ids):
    eng = orc.ConvolutionEngine()
    for p in range(4): eng.set_ir(p, irs[p])
    eq = orc.StereoParametricEQ(NB, synth.FS)
    for i in range(NB): eq.set_band_coeffs(i, coeffs[i], bool(en[i]))
    l, r = x[k, 0].copy(), x[k, 1].copy()
    orc.chain_process(eng, eq, l, r, eq_enable=True, gain=1.0)
    out[sid, 0], out[sid, 1] = l, r
t = torch.from_numpy(out)
dist.all_reduce(t)            # test-only gather of the disjoint shards (the data path has no collective)
if rank == 0:
    np.save({outfile!r}, t.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_world_size_2_gloo_sharded_run_matches_single_process(tmp_path, oracle):
    """N > 1 plan on CPU: rank 0 broadcasts the HRIR set + EQ table, each rank processes its
    contiguous shard of stream ids, and the union equals the single-process result."""
    from open_headstage_amd import synth
    outfile = str(tmp_path / "sharded.npy")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, outfile=outfile))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    got = np.load(outfile)
    TAPS, NB, S, FRAMES = 128, 10, 6, 2048
    irs = synth.hrir_set(TAPS)
    bands = synth.eq_table()
    x = synth.white_noise(range(S), FRAMES)
    for sid in range(S):
        eng = oracle.ConvolutionEngine()
        for p in range(4):
            eng.set_ir(p, irs[p])
        eq = oracle.StereoParametricEQ(NB, synth.FS)
        for i, b in enumerate(bands):
            eq.set_band_coeffs(i, oracle.biquad_coeffs(int(b.filter_type), synth.FS, b.center_freq, b.q, b.gain_db), b.enabled)
        l, r = x[sid, 0].copy(), x[sid, 1].copy()
        oracle.chain_process(eng, eq, l, r, eq_enable=True, gain=1.0)
        assert np.array_equal(got[sid, 0], l) and np.array_equal(got[sid, 1], r), sid


def test_header_is_plain_c_and_the_c_host_links(tmp_path):
    """include/ohs_hip.h must be consumable by a C compiler (the Rust side binds a C ABI), and
    examples/host_chain.c must link against the built library (no GPU needed for either)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    libdir = os.path.join(root, "open_headstage_amd")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"), "-o",
                    str(tmp_path / "host_chain"), os.path.join(root, "examples", "host_chain.c"),
                    "-L" + libdir, "-lohs_hip", "-lm", "-Wl,-rpath," + libdir], check=True)


def test_kernel_register_budgets(lib):
    """Figures hipcc reported when the library was built (open_headstage_amd/libohs_hip.resources.json).
    k_eq_ring must fit beside four k_conv_p1 waves on a SIMD: 4 x alloc(k_conv_p1) + alloc(k_eq_ring) <= 512
    registers per lane at the hardware's granule of 8, and neither may spill."""
    from open_headstage_amd import build
    res = build.resources()
    if not res:
        pytest.skip("library built before resource figures were recorded")

    def alloc(k):
        return -(-(res[k]["vgprs"] + res[k]["agprs"]) // 8) * 8

    assert res["k_eq_ring"]["scratch_bytes_per_lane"] == 0
    assert res["k_conv_p1"]["scratch_bytes_per_lane"] == 0
    # the single-stream kernels: nothing in scratch memory on a real-time call's path (round 3: 92 B per lane in the
    # resident kernel -- thread 0's time stamps and loop-invariant addresses alive across the call body)
    for k in ("k_engine_resident", "k_engine_call", "k_engine_presum", "k_conv_general", "k_conv_p1_edges", "k_conv_p1_state", "k_conv_tp_old",
              "k_conv_tp_mac<16>", "k_conv_tp_mac<32>", "k_conv_tp_forward", "k_conv_tp_inverse", "k_os_tables"):
        assert res[k]["scratch_bytes_per_lane"] == 0, (k, res[k])
    assert res["k_conv_general"]["vgprs"] <= 256 and res["k_conv_general"]["agprs"] <= 64      # (AGPR moves, no scratch: a fallback)
    assert alloc("k_eq_ring") <= 32, res["k_eq_ring"]
    assert 4 * alloc("k_conv_p1") + alloc("k_eq_ring") <= 512, (res["k_conv_p1"], res["k_eq_ring"])
    # the generated asm bodies must be the generator's current output
    import importlib.util
    gen_path = os.path.join(ROOT, "tools", "gen_eq_ring2_asm.py")
    spec = importlib.util.spec_from_file_location("gen_eq_ring2_asm", gen_path)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    inc = open(os.path.join(ROOT, "open_headstage_amd", "csrc", "eq_ring2_asm.inc")).read()
    for macro, lines in (("EQ_RING2_MAIN", gen.gen_main()),):
        body = f"#define {macro} \\\n" + " \\\n".join('    "' + l + '\\n"' for l in lines) + "\n"
        assert body in inc, f"{macro}: eq_ring2_asm.inc is not what tools/gen_eq_ring2_asm.py writes"


def test_bench_self_launch_plumbing_without_gpu():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent starts two ranks of itself and returns the worst
    child status.  Without a GPU every rank refuses (exit 2, no CPU fallback) -- which exercises the launcher, the
    environment it builds and the status relay here; the GPU rehearsal is tests/test_gpu_bench_rehearsal.py."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here: covered by the GPU rehearsal")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-2000:])
    assert r.stderr.count("no GPU visible") == 2, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # a job with fewer streams than ranks fails on every rank before any process group exists
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--total-streams", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "a rank would have no stream" in r.stderr


def test_the_product_library_has_no_debug_entry_points_and_reads_no_environment():
    import ctypes as C
    import os
    import subprocess
    from open_headstage_amd import _ffi
    prod = C.CDLL(os.path.join(_ffi.HERE, "libohs_hip.so"))
    for name in _ffi.DEBUG_PROTOTYPES:
        assert not hasattr(prod, name), name
    # the only strings of the product library that start with OHS_ are the status names of ohs_status_string
    out = subprocess.run(["strings", os.path.join(_ffi.HERE, "libohs_hip.so")], capture_output=True, text=True).stdout
    names = sorted({l for l in out.splitlines() if l.startswith("OHS_")})
    assert names and all(n == "OHS_OK" or n.startswith("OHS_ERR_") for n in names), names
    assert "getenv" not in subprocess.run(["nm", "-D", "--undefined-only", os.path.join(_ffi.HERE, "libohs_hip.so")],
                                          capture_output=True, text=True).stdout


def test_the_hop_kernels_steady_loop_does_not_spill(tmp_path):
    """k_conv_p1_os runs three waves per SIMD (168 registers per lane).  Left alone hipcc over-commits its product phase
    (130 scratch instructions per hop: 0.92 instead of 0.49 ms per headline-sized step); the kernel parks the results of
    slots {0, 1} in the wave's idle exchange slab to cut those live ranges.  Held here on the ISA of the steady hop --
    the kernel-wide scratch figure also covers the guarded first / last hops, which may spill."""
    import re
    import shutil
    from open_headstage_amd import build
    res = build.resources()
    if res:
        assert res["k_conv_p1_os"]["occupancy_waves_per_simd"] == 3 and res["k_conv_p1_os"]["vgprs"] <= 168, res["k_conv_p1_os"]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    csrc = os.path.join(ROOT, "open_headstage_amd", "csrc")
    out = tmp_path / "os.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-ffp-contract=off", "-S",
                    "--cuda-device-only", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-o", str(out),
                    os.path.join(csrc, "conv_os_kernels.hip")], check=True, timeout=600)
    lines = out.read_text().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*k_conv_p1_os\w*:", l))
    end = next(i for i, l in enumerate(lines) if i > start and ".Lfunc_end" in l)
    lines = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and 2000 < i - labels[m.group(1)]:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    assert loops, "no hop loop found in k_conv_p1_os"
    _, a, b = min(loops)                        # the innermost big loop: the steady hops
    body = [x.strip().split()[0] for x in lines[a:b] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
    stores = sum(1 for op in body if op.startswith("scratch_store"))
    loads = sum(1 for op in body if op.startswith("scratch_load"))
    assert stores == 0 and loads <= 8, f"steady hop: {stores} scratch stores, {loads} scratch loads in {len(body)} instructions"
    assert 2800 < len(body) < 3700, len(body)
    # The WHOLE kernel: the 260 B / lane of scratch the resource report shows belong to the guarded
    # hops -- the first hop of a range (state in: merged or per-path overlaps, zero history in front of the call) in front
    # of the steady loop, and the loop behind it that runs a range's last one or two hops (ragged end, state out: merged
    # overlap + last input block) -- plus values hipcc parks across the steady loop for them.  Each of those runs once or
    # twice per wave and launch against ~80 steady hops of ~3 250 instructions (headline shape): < 0.1 % of the
    # instructions issued.  Held here so that a change that lets scratch grow -- or creep into the loop -- shows.
    def scratch(lo, hi):
        ops = [x.strip().split()[0] for x in lines[lo:hi] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
        return sum(1 for op in ops if op.startswith("scratch_store")), sum(1 for op in ops if op.startswith("scratch_load"))
    before, after = scratch(0, a), scratch(b, len(lines))
    print(f"k_conv_p1_os scratch instructions (stores, loads): in front of the steady loop {before}, inside {(stores, loads)}, "
          f"behind it {after}")
    assert before[0] <= 60 and before[1] <= 40, before
    assert after[0] <= 80 and after[1] <= 80, after
    if res:
        assert res["k_conv_p1_os"]["scratch_bytes_per_lane"] <= 272, res["k_conv_p1_os"]
