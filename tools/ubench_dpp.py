#!/usr/bin/env python3
"""Lone-wave issue cost and dependent latency of the instruction kinds the EQ kernels use (gfx950).
Writes + compiles tools/bin/ubench_dpp;   gpurun -- tools/bin/ubench_dpp
Each case is a 64x unrolled asm block run 2000 times by one wave; result = s_memtime ticks per instruction."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
FULL = "row_mask:0xf bank_mask:0xf"
CASES = {}


def case(name, lines):
    CASES[name] = lines


def indep(fmt, n=64):      # fmt uses {d} = destination register index, independent sources
    return [fmt.format(d=20 + 2 * (i % 8)) for i in range(n)]


case("indep v_add_f32", indep("v_add_f32 v{d}, v2, v3"))
case("indep v_mov_b32_dpp ror1", indep("v_mov_b32_dpp v{d}, v2 row_ror:1 " + FULL))
case("indep v_add_f32_dpp ror1", indep("v_add_f32_dpp v{d}, v2, v3 row_ror:1 " + FULL))
case("indep v_mul_f32_dpp ror1", indep("v_mul_f32_dpp v{d}, v2, v3 row_ror:1 " + FULL))
case("indep v_add_f32_dpp quad identity", indep("v_add_f32_dpp v{d}, v2, v3 quad_perm:[0,1,2,3] " + FULL))
case("indep v_pk_mul_f32", indep("v_pk_mul_f32 v[{d}:{d}+1], v[2:3], v[4:5]".replace("{d}+1", "{e}")) if False else
     [f"v_pk_mul_f32 v[{20 + 2 * (i % 8)}:{21 + 2 * (i % 8)}], v[2:3], v[4:5]" for i in range(64)])
case("indep v_pk_add_f32", [f"v_pk_add_f32 v[{20 + 2 * (i % 8)}:{21 + 2 * (i % 8)}], v[2:3], v[4:5]" for i in range(64)])
# dependent chains (each instruction reads the previous result)
case("dep v_add_f32", ["v_add_f32 v20, v20, v3"] * 64)
case("dep v_pk_add_f32", ["v_pk_add_f32 v[20:21], v[20:21], v[4:5]"] * 64)
case("dep v_pk_mul_f32", ["v_pk_mul_f32 v[20:21], v[20:21], v[4:5]"] * 64)
case("dep v_add_f32 via src1 of dpp op (+2 indep)", ["v_add_f32_dpp v20, v2, v20 row_ror:1 " + FULL, "v_add_f32 v30, v2, v3", "v_add_f32 v32, v2, v3"] * 21)
case("dep dpp-src chain (+2 indep between)", ["v_add_f32_dpp v20, v20, v3 row_ror:1 " + FULL, "v_add_f32 v30, v2, v3", "v_add_f32 v32, v2, v3"] * 21)
case("dep dpp-src chain (+3 indep between)", ["v_add_f32_dpp v20, v20, v3 row_ror:1 " + FULL, "v_add_f32 v30, v2, v3", "v_add_f32 v32, v2, v3", "v_add_f32 v34, v2, v3"] * 16)
case("dep mov_dpp chain (+2 indep between)", ["v_mov_b32_dpp v20, v20 row_ror:1 " + FULL, "v_add_f32 v30, v2, v3", "v_add_f32 v32, v2, v3"] * 21)
# the three-op recurrence of the ring body: O (add_dpp) -> A (pk_mul) -> N (pk_add) -> O, one indep between each
case("loop O A N, 1 indep between", ["v_add_f32_dpp v20, v26, v22 row_ror:1 " + FULL, "v_add_f32 v30, v2, v3",
                                      "v_pk_mul_f32 v[24:25], v[4:5], v[20:21] op_sel_hi:[1,0]", "v_add_f32 v32, v2, v3",
                                      "v_pk_add_f32 v[22:23], v[6:7], v[24:25] neg_lo:[0,1] neg_hi:[0,1]", "v_add_f32 v34, v2, v3"] * 10)
case("loop O A N (plain add O), 1 indep between", ["v_add_f32 v20, v26, v22", "v_add_f32 v30, v2, v3",
                                      "v_pk_mul_f32 v[24:25], v[4:5], v[20:21] op_sel_hi:[1,0]", "v_add_f32 v32, v2, v3",
                                      "v_pk_add_f32 v[22:23], v[6:7], v[24:25] neg_lo:[0,1] neg_hi:[0,1]", "v_add_f32 v34, v2, v3"] * 10)
case("loop O A N (plain), no filler", ["v_add_f32 v20, v26, v22",
                                      "v_pk_mul_f32 v[24:25], v[4:5], v[20:21] op_sel_hi:[1,0]",
                                      "v_pk_add_f32 v[22:23], v[6:7], v[24:25] neg_lo:[0,1] neg_hi:[0,1]"] * 21)
case("loop O A N (plain), scalar mul A", ["v_add_f32 v20, v26, v22", "v_add_f32 v30, v2, v3",
                                      "v_mul_f32 v24, v4, v20", "v_add_f32 v32, v2, v3",
                                      "v_sub_f32 v22, v6, v24", "v_add_f32 v34, v2, v3"] * 10)


def main():
    src = ["#include <hip/hip_runtime.h>", "#include <cstdio>"]
    names = list(CASES)
    for i, n in enumerate(names):
        body = "\\n".join(CASES[n]) + "\\n"
        clob = ", ".join(f'"v{r}"' for r in range(20, 36))
        src.append(f"""__global__ void k{i}(float* buf, unsigned long long* res, int iters)
{{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
        asm volatile(".p2align 5\\nv_mov_b32 v2, 1.0\\nv_mov_b32 v3, 0.5\\nv_mov_b32 v4, 0.5\\nv_mov_b32 v5, 0.25\\nv_mov_b32 v6, 0.5\\nv_mov_b32 v7, 0.25\\ns_nop 4\\n{body}" ::: "v2","v3","v4","v5","v6","v7",{clob},"memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
}}""")
    src.append("int main() {\n    float* buf; unsigned long long* res; hipMalloc(&buf, 1 << 20); hipMalloc(&res, 64);\n    const int iters = 2000; unsigned long long h;")
    for i, n in enumerate(names):
        cnt = len(CASES[n])
        src.append(f'    for (int r = 0; r < 2; ++r) {{ hipLaunchKernelGGL(k{i}, dim3(1), dim3(64), 0, 0, buf, res, iters); hipDeviceSynchronize(); }}\n'
                   f'    hipMemcpy(&h, res, 8, hipMemcpyDeviceToHost); printf("%-52s %6.2f ticks / instruction  (%d per block, %.1f per block)\\n", "{n}", (double)h / (iters * {cnt}.0), {cnt}, (double)h / iters);')
    src.append("    return 0;\n}")
    open("/tmp/ubench_dpp.hip", "w").write("\n".join(src))
    os.makedirs(os.path.join(HERE, "bin"), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-w", "--offload-arch=gfx950", "-o", os.path.join(HERE, "bin", "ubench_dpp"), "/tmp/ubench_dpp.hip"], check=True)


if __name__ == "__main__":
    main()
