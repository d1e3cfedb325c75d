"""Builds libohs_hip.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m open_headstage_amd.build [--force]

hipcc cross-compiles without a GPU.  eq_kernels.hip is compiled with
-ffp-contract=off: the biquad recurrence must round every product and sum
separately to be bit-exact with the reference's arithmetic.

Staleness is decided by CONTENT (sha256 of each unit's source, every header / .inc it may include, its
flags and this file), recorded in libohs_hip.stamp next to the library -- not by mtimes, which a
snapshot copy to the GPU box does not keep.  `is_current()` lets the test session refuse a library
that does not match the sources it is about to test.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys

# (TAG / EXPERIMENTS below read sys.argv: `python -m open_headstage_amd.build --experiments`)
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# Experiment variants: OHS_BUILD_TAG=name OHS_EXTRA_DEFS="-DFOO=1 ..." builds libohs_hip_name.so beside the
# product library (objects under build/name/); OHS_LIB=<path> makes _ffi.py load it.  Not used in production.
#
# Two libraries come out of this file:
#   libohs_hip.so       the PRODUCT: reads no environment variable, no experiment variant compiled in (csrc/experiments.h)
#   libohs_hip_exp.so   the EXPERIMENTS build (-DOHS_EXPERIMENTS): the same code plus the OHS_* environment knobs, the debug
#                       entry points (ohs_debug_set_tuning ...) and round 2's EQ kernel; `--experiments`, or
#                       OHS_BUILD_TAG=exp.  Other tags (OHS_BUILD_TAG=name OHS_EXTRA_DEFS="-DFOO") are variants OF the
#                       experiments build: every tagged library is compiled with -DOHS_EXPERIMENTS.
TAG = os.environ.get("OHS_BUILD_TAG", "")
if "--experiments" in sys.argv and not TAG:
    TAG = "exp"
EXPERIMENTS = bool(TAG)
EXTRA_DEFS = os.environ.get("OHS_EXTRA_DEFS", "").split() + (["-DOHS_EXPERIMENTS=1"] if EXPERIMENTS else [])
OBJ = os.path.join(HERE, "build", TAG) if TAG else os.path.join(HERE, "build")
LIB = os.path.join(HERE, f"libohs_hip_{TAG}.so" if TAG else "libohs_hip.so")
STAMP = os.path.join(HERE, f"libohs_hip_{TAG}.stamp" if TAG else "libohs_hip.stamp")
# per-kernel register / scratch / LDS figures as hipcc reports them (-Rpass-analysis=kernel-resource-usage), written
# next to the library at build time: DESIGN.md quotes them and tests/test_cpu_host_logic.py holds k_eq_ring to its
# 32-VGPR budget with them
RESOURCES = os.path.join(HERE, f"libohs_hip_{TAG}.resources.json" if TAG else "libohs_hip.resources.json")
ARCH = "gfx950"


def _units():
    # (source, extra flags)
    return [
        # FFT kernels: no SLP vectoriser (see conv_mac_kernels.hip); no implicit FMA contraction -- the fused
        # operations are written out (wave_fft.hpp), so that every kernel inlining them rounds alike
        ("conv_kernels.hip", ["-fno-slp-vectorize", "-ffp-contract=off"]),
        ("conv_os_kernels.hip", ["-fno-slp-vectorize", "-ffp-contract=off"]),   # the overlap-save plan of the P = 1 path
        ("conv_lb_kernels.hip", ["-fno-slp-vectorize", "-ffp-contract=off"]),   # block 2048 / FFT 4096 for long impulse responses
        ("conv_xb_kernels.hip", ["-fno-slp-vectorize", "-ffp-contract=off"]),   # block 8192 / FFT 16384 in one kernel: their long out-of-place calls
        # (experiments builds only: OHS_MAC_TB / OHS_MAC_PI fix the MAC's register tile)
        ("conv_mac_kernels.hip", [f"-D{k}={os.environ[k]}" for k in ("OHS_MAC_TB", "OHS_MAC_PI")
                                  if EXPERIMENTS and k in os.environ]),
        ("eq_kernels.hip", ["-ffp-contract=off"]),
        ("api_core.hip", []), ("api_conv.hip", []), ("api_eq.hip", []), ("api_engine.hip", []), ("api_batch.hip", []),   # the C ABI (api_internal.h)
        ("sofa_reader.cpp", ["-x", "hip"]),     # host-only C++ (HDF5 subset reader), built by the same driver
        ("sofa_conditioning.cpp", ["-x", "hip"]),   # host-only C++ (libmysofa-style loudness / interpolation)
        ("autoeq_parser.cpp", ["-x", "hip"]),   # host-only C++ (AutoEQ CSV)
        ("node_batch.cpp", ["-x", "hip"]),      # host-only C++ (batch mode over the GPUs of a node; RCCL by dlopen)
        ("speakers.cpp", ["-x", "hip"]),        # host-only C++ (speaker angles -> four HRIRs off a SOFA handle)
        ("tuning.cpp", ["-x", "hip"]),          # host-only C++ (launch-plan constants; environment knobs in experiments builds)
        # host-only C++ (RBJ coefficient formulas + the musl restatement of sinf / cosf / powf: every operation rounds
        # by itself)
        ("biquad_coeffs.cpp", ["-x", "hip", "-ffp-contract=off"]),
    ] + ([
        # experiments builds only: the block-8192 plan's first form (sixteen waves, pair image) -- parity-green, slower (LABNOTES.md)
        ("experiments/conv_xb16_kernels.hip", ["-fno-slp-vectorize", "-ffp-contract=off"]),
    ] if EXPERIMENTS else [])


COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _deps(src: str) -> list[str]:
    deps = [os.path.join(CSRC, src)]
    deps += sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".inc")))
    exp_dir = os.path.join(CSRC, "experiments")         # (included under -DOHS_EXPERIMENTS only)
    if os.path.isdir(exp_dir):
        deps += sorted(os.path.join(exp_dir, f) for f in os.listdir(exp_dir) if f.endswith((".h", ".hpp", ".inc")))
    deps.append(os.path.join(os.path.dirname(HERE), "include", "ohs_hip.h"))
    deps.append(os.path.join(CSRC, "exports.map"))
    deps.append(os.path.abspath(__file__))
    return deps


def _unit_hash(src: str, extra: list[str]) -> str:
    h = hashlib.sha256()
    h.update(" ".join(COMMON + extra + EXTRA_DEFS).encode())
    for d in _deps(src):
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _current_hashes() -> dict:
    return {src: _unit_hash(src, extra) for src, extra in _units()}


def _read_stamp() -> dict:
    try:
        with open(STAMP) as f:
            return json.load(f)
    except Exception:
        return {}


def is_current() -> bool:
    """True when libohs_hip.so exists and was built from exactly the sources in the tree."""
    return os.path.exists(LIB) and _read_stamp() == _current_hashes()


def build(force: bool = False, verbose: bool = False) -> str:
    units = _units()
    missing = [s for s, _ in units if not os.path.exists(os.path.join(CSRC, s))]
    if missing:
        raise RuntimeError(f"source files missing under {CSRC}: {missing}")
    want = _current_hashes()
    have = _read_stamp()
    if not force and os.path.exists(LIB) and have == want:
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for src, extra in units:
        obj = os.path.join(OBJ, os.path.splitext(src)[0].replace("/", "_") + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or _read_obj_hash(obj) != want[src]:
            cmd = [hipcc, *COMMON, *extra, *EXTRA_DEFS, "-Rpass-analysis=kernel-resource-usage", "-fno-caret-diagnostics", "-c",
                   os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            remarks = [l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" in l]
            rest = [l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l]
            if rest:
                sys.stderr.write("\n".join(rest) + "\n")
            if r.returncode != 0:
                raise subprocess.CalledProcessError(r.returncode, cmd)
            with open(obj + ".resources.json", "w") as f:
                json.dump(_parse_resource_remarks(remarks), f, indent=1, sort_keys=True)
            with open(obj + ".hash", "w") as f:
                f.write(want[src])
    # exported surface = the C ABI (csrc/exports.map); the C++ internals stay local
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs,
           "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-lz", "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    res = {}
    for obj in objs:
        try:
            with open(obj + ".resources.json") as f:
                res.update(json.load(f))
        except Exception:       # noqa: BLE001 -- an object built before this file recorded them
            pass
    with open(RESOURCES, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    with open(STAMP, "w") as f:
        json.dump(want, f, indent=1)
    return LIB


def _parse_resource_remarks(lines: list[str]) -> dict:
    """{kernel (demangled base name, template arguments kept): {vgprs, agprs, sgprs, scratch_bytes_per_lane,
    lds_bytes, occupancy_waves_per_simd}} from hipcc's kernel-resource-usage remarks."""
    import re
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
            "LDS Size [bytes/block]": "lds_bytes", "Occupancy [waves/SIMD]": "occupancy_waves_per_simd"}
    for l in lines:
        m = re.search(r"remark:\s+Function Name: (\S+)", l)
        if m:
            name = m.group(1)
            name = _kernel_base_name(name)
            cur = out.setdefault(name, {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", l)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def _kernel_base_name(mangled: str) -> str:
    """_ZN3ohs9k_conv_p1ENS_10ConvP1ArgsE -> k_conv_p1; _ZN3ohs9k_eq_passILi6EEEv... -> k_eq_pass<6> (Itanium
    nested-name <length><identifier> pairs; no demangler is needed for the kernels of this library)"""
    import re
    m = re.match(r"_ZN3ohs(\d+)", mangled)
    if not m:
        return mangled
    n = int(m.group(1))
    start = m.end()
    name = mangled[start:start + n]
    t = re.match(r"ILi(\d+)E", mangled[start + n:])
    return f"{name}<{t.group(1)}>" if t else name


def resources() -> dict:
    """the figures recorded by the last build (empty if the library was built before they were recorded)"""
    try:
        with open(RESOURCES) as f:
            return json.load(f)
    except Exception:           # noqa: BLE001
        return {}


def _read_obj_hash(obj: str) -> str:
    try:
        with open(obj + ".hash") as f:
            return f.read().strip()
    except Exception:
        return ""


def build_experiments(force: bool = False) -> str:
    """libohs_hip_exp.so beside the product library (a child process: this module's paths are fixed at import)"""
    env = dict(os.environ, OHS_BUILD_TAG="exp")
    env.pop("OHS_EXTRA_DEFS", None)
    cmd = [sys.executable, "-m", "open_headstage_amd.build"] + (["--force"] if force else [])
    subprocess.run(cmd, check=True, cwd=os.path.dirname(HERE), env=env, stdout=subprocess.DEVNULL)
    return os.path.join(HERE, "libohs_hip_exp.so")


def experiments_is_current() -> bool:
    r = subprocess.run([sys.executable, "-c", "import sys; from open_headstage_amd import build as b; sys.exit(0 if b.is_current() else 1)"],
                       cwd=os.path.dirname(HERE), env=dict(os.environ, OHS_BUILD_TAG="exp"))
    return r.returncode == 0


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
