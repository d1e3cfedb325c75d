#!/usr/bin/env python3
"""Hidden stalls between a VALU write and a later read on a lone gfx950 wave: producer, k independent
filler instructions, consumer -- cycles per unit minus 4.1 x (k + 2) is the stall.
Writes + compiles tools/bin/ubench_hazard;   gpurun -- tools/bin/ubench_hazard"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
FULL = "row_mask:0xf bank_mask:0xf"
PROD = {
    "add": "v_add_f32 v20, v2, v3",
    "pk_mul": "v_pk_mul_f32 v[20:21], v[2:3], v[4:5]",
    "add_dpp": "v_add_f32_dpp v20, v2, v3 row_ror:1 " + FULL,
    "mov_dpp": "v_mov_b32_dpp v20, v2 row_ror:1 " + FULL,
}
CONS = {
    "add": "v_add_f32 v30, v20, v3",
    "pk_mul": "v_pk_mul_f32 v[30:31], v[20:21], v[4:5]",
    "pk_mul(bcast lo)": "v_pk_mul_f32 v[30:31], v[4:5], v[20:21] op_sel_hi:[1,0]",
    "add_dpp(dpp src)": "v_add_f32_dpp v30, v20, v3 row_ror:1 " + FULL,
    "add_dpp(src1)": "v_add_f32_dpp v30, v3, v20 row_ror:1 " + FULL,
    "mov_dpp": "v_mov_b32_dpp v30, v20 row_ror:1 " + FULL,
}
FILL = ["v_add_f32 v24, v2, v3", "v_add_f32 v25, v2, v3", "v_add_f32 v26, v2, v3", "v_add_f32 v27, v2, v3", "v_add_f32 v28, v2, v3"]
UNITS = 64


def main():
    src = ["#include <hip/hip_runtime.h>", "#include <cstdio>"]
    cases = []
    for pn, p in PROD.items():
        for cn, c in CONS.items():
            for k in range(0, 5):
                if "dpp" in cn and cn != "add_dpp(src1)" and k < 2:
                    continue            # architectural hazard: needs 2 wait states
                cases.append((pn, cn, k, [p] + FILL[:k] + [c]))
    for i, (pn, cn, k, unit) in enumerate(cases):
        body = "\\n".join(unit * UNITS) + "\\n"
        clob = ", ".join(f'"v{r}"' for r in range(20, 32))
        src.append(f"""__global__ void k{i}(unsigned long long* res, int iters)
{{
    asm volatile("v_mov_b32 v2, 1.0\\nv_mov_b32 v3, 0.5\\nv_mov_b32 v4, 0.5\\nv_mov_b32 v5, 0.25\\nv_mov_b32 v20, 0\\nv_mov_b32 v21, 0" ::: "v2","v3","v4","v5","v20","v21");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
        asm volatile(".p2align 6\\n{body}" ::: "v2","v3","v4","v5",{clob},"memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
}}""")
    src.append("int main() {\n    unsigned long long* res; hipMalloc(&res, 1024);\n    const int iters = 300; unsigned long long h;")
    last = None
    for i, (pn, cn, k, unit) in enumerate(cases):
        if (pn, cn) != last:
            src.append(f'    printf("\\n%-10s -> %-18s stall cycles at k fillers:", "{pn}", "{cn}");')
            last = (pn, cn)
        src.append(f'    for (int r = 0; r < 2; ++r) {{ hipLaunchKernelGGL(k{i}, dim3(1), dim3(64), 0, 0, res, iters); hipDeviceSynchronize(); }}\n'
                   f'    hipMemcpy(&h, res, 8, hipMemcpyDeviceToHost); printf("  k={k}: %5.1f", (double)h / (iters * {UNITS}.0) - 4.1 * {k + 2});')
    src.append('    printf("\\n"); return 0;\n}')
    open("/tmp/ubench_hazard.hip", "w").write("\n".join(src))
    os.makedirs(os.path.join(HERE, "bin"), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-w", "--offload-arch=gfx950", "-o", os.path.join(HERE, "bin", "ubench_hazard"), "/tmp/ubench_hazard.hip"], check=True)


if __name__ == "__main__":
    main()
