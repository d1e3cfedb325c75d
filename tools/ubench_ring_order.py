#!/usr/bin/env python3
"""Issue-order sweep for the EQ ring group body (tools/gen_eq_ring_asm.py) in isolation: one wave,
cycles per sample for every hazard-free order of the six per-step tokens (plus a few NOP-padded ones).

    python tools/ubench_ring_order.py        # writes + compiles tools/bin/ubench_ring
    gpurun -- tools/bin/ubench_ring
"""
import itertools
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GEN = os.path.join(HERE, "gen_eq_ring_asm.py")


def valid(order):
    """order: tuple of tokens for one period, O T A N of step k and P M of step k+1."""
    pos = {t: i for i, t in enumerate(order) if t != "NOP"}
    n = len(order)
    before = lambda a, b: pos[a] < pos[b]
    if not (before("O", "A") and before("A", "N") and before("T", "N") and before("T", "P") and before("O", "P")
            and before("N", "M") and before("O", "M")):
        return False
    gap = lambda w, r: (pos[r] - pos[w] - 1) % n          # instructions between writer and (later) reader
    # DPP reads: O,T read P's result (previous period), M reads O's result
    return gap("P", "O") >= 2 and gap("P", "T") >= 2 and gap("O", "M") >= 2


def main():
    toks = ["O", "T", "A", "P", "N", "M"]
    orders = [o for o in itertools.permutations(toks) if o[0] == "O" and valid(o)]
    for extra in (["O", "T", "A", "P", "NOP", "N", "M"], ["O", "T", "NOP", "A", "P", "N", "M"],
                  ["O", "NOP", "T", "A", "P", "N", "M"], ["O", "T", "A", "NOP", "P", "N", "NOP", "M"]):
        if valid(tuple(extra)):
            orders.append(tuple(extra))
    src = ["#include <hip/hip_runtime.h>", "#include <cstdio>", "typedef float v2f __attribute__((ext_vector_type(2)));"]
    knock = [""] * len(orders)
    base = ("O", "T", "A", "P", "N", "M")
    for kn in ("store", "load", "port", "store load", "store load port"):
        orders.append(base)
        knock.append(kn)
    for i, o in enumerate(orders):
        inc = f"/tmp/ring_v{i}.inc"
        env = dict(os.environ, EQ_RING_ORDER=" ".join(o), EQ_RING_KNOCKOUT=knock[i])
        subprocess.run([sys.executable, GEN, inc, f"BODY_v{i}"], check=True, env=env, stdout=subprocess.DEVNULL)
        src.append(f'#include "{inc}"')
        src.append(f"""__global__ void k_v{i}(float* buf, unsigned long long* res, int iters, float c0, float c1)
{{
    const int lane = threadIdx.x;
    float s1 = 0.f, s2 = 0.f, o = lane * 1e-3f;
    v2f pb01 = {{c0, c1}}, a12 = {{c1, c0 * 0.5f}};
    float b2 = c1 * 0.3f;
    float x[8], nx[8];
    for (int e = 0; e < 8; ++e) x[e] = 0.01f * (e + lane);
    float* sp = buf + 4096 + (lane >> 4) * 256 + (lane & 15);
    const float* lp = buf + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {{
        asm volatile(BODY_v{i}
            : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [n0] "=&v"(nx[0]), [n1] "=&v"(nx[1]), [n2] "=&v"(nx[2]), [n3] "=&v"(nx[3]),
              [n4] "=&v"(nx[4]), [n5] "=&v"(nx[5]), [n6] "=&v"(nx[6]), [n7] "=&v"(nx[7])
            : [pb01] "v"(pb01), [b2] "v"(b2), [a12] "v"(a12), [sp] "v"(sp), [lp] "v"(lp), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]),
              [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7])
            : "v100","v102","v103","v106","v107","v108","v109","v110","v111","v112","v113","memory");
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]) :: "memory");
        for (int e = 0; e < 8; ++e) x[e] = nx[e];
    }}
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) res[0] = t1 - t0;
    buf[8192 + lane] = s1 + s2 + o;
}}""")
    src.append("int main() {\n    float* buf; unsigned long long* res; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20); hipMalloc(&res, 64);\n"
               "    const int iters = 2000; unsigned long long h;")
    for i, o in enumerate(orders):
        src.append(f'    for (int r = 0; r < 2; ++r) {{ hipLaunchKernelGGL(k_v{i}, dim3(1), dim3(64), 0, 0, buf, res, iters, 0.5f, 0.25f); hipDeviceSynchronize(); }}\n'
                   f'    hipMemcpy(&h, res, 8, hipMemcpyDeviceToHost); printf("%-40s %.2f memtime ticks / sample\\n", "{" ".join(o)} {("-" + knock[i]) if knock[i] else ""}", (double)h / (iters * 128.0));')
    src.append("    return 0;\n}")
    path = "/tmp/ubench_ring.hip"
    open(path, "w").write("\n".join(src))
    os.makedirs(os.path.join(HERE, "bin"), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-w", "--offload-arch=gfx950", "-o", os.path.join(HERE, "bin", "ubench_ring"), path], check=True)
    print(len(orders), "orders")


if __name__ == "__main__":
    main()
