"""CPU suite: the host-side parsers of untrusted files (SOFA / HDF5 subset, AutoEQ CSV) built with
AddressSanitizer + UBSan (gcc, CPU only) and run over regression inputs and a short mutation fuzz
(tools/fuzz_host_parsers.cpp).  The regression inputs are single-field patches of the reference's DATA asset
data/hrtf/subject_003.sofa (a data file; `need_asset` skips where it is absent):
  * Data.Delay dataspace [1][2] -> [2][2] / [0][2]: the round-1 reader indexed values[m*R + r] for every
    m < M (heap over-read, advisor finding);
  * a chunk B-tree node pointing at itself with level > 0 (unbounded walk);
  * a dataspace of 2^31 x 2^31 elements (allocation of more than the file can back)."""
import os
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOFA = "/root/reference/data/hrtf/subject_003.sofa"
need_asset = pytest.mark.skipif(not os.path.exists(SOFA), reason="reference data asset not present on this box")
DELAY_DIMS_OFF = 8256          # Data.Delay dataspace message: version 2, rank 2, dims u64 (1, 2) at this offset


@pytest.fixture(scope="module")
def fuzz_bin(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    out = str(tmp_path_factory.mktemp("asan") / "ohs_fuzz")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "open_headstage_amd", "csrc"),
           "-o", out, os.path.join(ROOT, "tools", "fuzz_host_parsers.cpp"),
           os.path.join(ROOT, "open_headstage_amd", "csrc", "sofa_reader.cpp"),
           os.path.join(ROOT, "open_headstage_amd", "csrc", "sofa_conditioning.cpp"),
           os.path.join(ROOT, "open_headstage_amd", "csrc", "autoeq_parser.cpp"), "-lz"]
    cmd = [c for c in cmd if not c.endswith(".cpp") or os.path.exists(c)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower() and "cannot find" in r.stderr:
        pytest.skip("libasan not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    return out


def _run(binary, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([binary, *args], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, \
        (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    return r.stdout


@need_asset
def test_pristine_asset_opens(fuzz_bin):
    assert "status 0" in _run(fuzz_bin, SOFA, "--open")


@need_asset
@pytest.mark.parametrize("dim0", [2, 0, 3, 1 << 40])
def test_data_delay_with_foreign_leading_dimension(fuzz_bin, tmp_path, dim0):
    raw = bytearray(open(SOFA, "rb").read())
    assert struct.unpack_from("<QQ", raw, DELAY_DIMS_OFF) == (1, 2)
    struct.pack_into("<Q", raw, DELAY_DIMS_OFF, dim0)
    p = tmp_path / "delay.sofa"
    p.write_bytes(raw)
    out = _run(fuzz_bin, str(p), "--open")       # either rejected or opened with the delays ignored: no over-read
    assert "open: status" in out


@need_asset
def test_self_referencing_btree_node_terminates(fuzz_bin, tmp_path):
    raw = bytearray(open(SOFA, "rb").read())
    hits = [i for i in range(len(raw) - 4) if raw[i:i + 4] == b"TREE" and raw[i + 4] == 1]
    assert hits
    for node in hits:
        # level := 1, entries used := 1, and the first child pointer := the node itself
        raw[node + 5] = 1
        struct.pack_into("<H", raw, node + 6, 1)
    p = tmp_path / "loop.sofa"
    # child address sits after the first key; key size depends on the rank, so try the ranks the file uses
    for rank in (1, 2, 3):
        r2 = bytearray(raw)
        for node in hits:
            struct.pack_into("<Q", r2, node + 8 + 16 + 8 + 8 * (rank + 1), node)
        p.write_bytes(r2)
        assert "open: status" in _run(fuzz_bin, str(p), "--open")


@need_asset
def test_huge_dataspace_is_rejected_not_allocated(fuzz_bin, tmp_path):
    raw = bytearray(open(SOFA, "rb").read())
    struct.pack_into("<QQ", raw, DELAY_DIMS_OFF, 1 << 31, 1 << 31)
    # the same on every rank-3 dataspace whose dims read (1250, 2, 200) = Data.IR
    pat = struct.pack("<QQQ", 1250, 2, 200)
    i = raw.find(pat)
    assert i > 0
    struct.pack_into("<QQQ", raw, i, 1 << 20, 1 << 20, 1 << 20)
    p = tmp_path / "huge.sofa"
    p.write_bytes(raw)
    out = _run(fuzz_bin, str(p), "--open")
    assert "status 8" in out or "status 6" in out        # OHS_ERR_PARSE (or ALLOC), never an abort


@need_asset
def test_short_mutation_fuzz(fuzz_bin):
    out = _run(fuzz_bin, SOFA, "160")
    assert "no crash" in out
