"""Builds libohs_hip.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m open_headstage_amd.build [--force]

hipcc cross-compiles without a GPU.  eq_kernels.hip is compiled with
-ffp-contract=off: the biquad recurrence must round every product and sum
separately to be bit-exact with the reference's arithmetic.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libohs_hip.so")
ARCH = "gfx950"

# (source, extra flags)
UNITS = [
    ("conv_kernels.hip", ["-fno-slp-vectorize"]),     # FFT kernels: see conv_mac_kernels.hip
    ("conv_mac_kernels.hip", [f"-DOHS_MAC_TB={os.environ.get('OHS_MAC_TB', '16')}",
                              f"-DOHS_MAC_PI={os.environ.get('OHS_MAC_PI', '16')}"]),
    ("eq_kernels.hip", ["-ffp-contract=off"]),
    ("ohs_api.hip", []),
    ("sofa_reader.cpp", ["-x", "hip"]),     # host-only C++ (HDF5 subset reader), built by the same driver
    ("autoeq_parser.cpp", ["-x", "hip"]),   # host-only C++ (AutoEQ CSV)
    ("biquad_coeffs.cpp", ["-x", "hip"]),   # host-only C++ (RBJ coefficient formulas)
]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _deps(src: str) -> list[str]:
    deps = [os.path.join(CSRC, src)]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".inc"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "ohs_hip.h"))
    deps.append(os.path.abspath(__file__))
    return deps


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    relink = force or not os.path.exists(LIB)
    for src, extra in UNITS:
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or any(
            os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps(src))
        if stale:
            cmd = [hipcc, *COMMON, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            relink = True
    if relink or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs, "-lz"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
