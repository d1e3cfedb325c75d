#!/usr/bin/env python3
"""Is a lone wave on gfx950 bound by instruction FETCH?  Straight-line blocks of 512 independent VALU
instructions of 4-byte (VOP2 e32) and 8-byte (VOP3 / DPP / VOP3P) encodings, 64-byte aligned, timed for
1, 2, 4, 8 and 16 waves per workgroup (16 = 4 per SIMD) -- per-wave s_memtime ticks per instruction.
Writes + compiles tools/bin/ubench_fetch;   gpurun -- tools/bin/ubench_fetch"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
FULL = "row_mask:0xf bank_mask:0xf"
N = 512
CASES = {
    "4B  v_add_f32 e32": ["v_add_f32 v{d}, v2, v3"],
    "8B  v_add_f32 e64": ["v_add_f32_e64 v{d}, v2, v3"],
    "8B  v_add_f32_dpp": ["v_add_f32_dpp v{d}, v2, v3 row_ror:1 " + FULL],
    "8B  v_pk_mul_f32": ["v_pk_mul_f32 v[{d}:{e}], v[2:3], v[4:5]"],
    "4B+8B alternating": ["v_add_f32 v{d}, v2, v3", "v_pk_mul_f32 v[{d}:{e}], v[2:3], v[4:5]"],
    "4B+4B+8B": ["v_add_f32 v{d}, v2, v3", "v_mul_f32 v{d}, v2, v3", "v_pk_mul_f32 v[{d}:{e}], v[2:3], v[4:5]"],
    "4B  s_nop 0": ["s_nop 0"],
    "4B dependent v_add_f32": ["v_add_f32 v20, v20, v3"],
    "8B dependent v_pk_add_f32": ["v_pk_add_f32 v[20:21], v[20:21], v[4:5]"],
    "8B dep dpp (+s_nop 1)": ["v_add_f32_dpp v20, v20, v3 row_ror:1 " + FULL, "s_nop 1"],
}


def main():
    src = ["#include <hip/hip_runtime.h>", "#include <cstdio>"]
    names = list(CASES)
    for i, n in enumerate(names):
        pat = CASES[n]
        lines = []
        k = 0
        while len(lines) < N:
            for p in pat:
                d = 20 + 2 * (k % 8)
                lines.append(p.format(d=d, e=d + 1))
                k += 1
        body = "\\n".join(lines) + "\\n"
        clob = ", ".join(f'"v{r}"' for r in range(20, 36))
        src.append(f"""__global__ void k{i}(unsigned long long* res, int iters)
{{
    asm volatile("v_mov_b32 v2, 1.0\\nv_mov_b32 v3, 0.5\\nv_mov_b32 v4, 0.5\\nv_mov_b32 v5, 0.25\\nv_mov_b32 v20, 0\\nv_mov_b32 v21, 0" ::: "v2","v3","v4","v5","v20","v21");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
        asm volatile(".p2align 6\\n{body}" ::: "v2","v3","v4","v5",{clob},"memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) res[threadIdx.x >> 6] = t1 - t0;
}}""")
    src.append("int main() {\n    unsigned long long* res; hipMalloc(&res, 1024);\n    const int iters = 500; unsigned long long h[16];")
    src.append('    printf("%-28s", "waves per workgroup:"); for (int w : {1, 2, 4, 8, 16}) printf(" %7d", w); printf("\\n");')
    for i, n in enumerate(names):
        cnt = len([0]) and N + (len(CASES[n]) - N % len(CASES[n])) % len(CASES[n])
        src.append(f'    printf("%-28s", "{n}");\n    for (int w : {{1, 2, 4, 8, 16}}) {{\n'
                   f'        for (int r = 0; r < 2; ++r) {{ hipLaunchKernelGGL(k{i}, dim3(1), dim3(64 * w), 0, 0, res, iters); hipDeviceSynchronize(); }}\n'
                   f'        hipMemcpy(h, res, 8 * w, hipMemcpyDeviceToHost); unsigned long long m = 0; for (int q = 0; q < w; ++q) m = h[q] > m ? h[q] : m;\n'
                   f'        printf(" %7.2f", (double)m / (iters * {cnt}.0)); }}\n    printf("\\n");')
    src.append("    return 0;\n}")
    open("/tmp/ubench_fetch.hip", "w").write("\n".join(src))
    os.makedirs(os.path.join(HERE, "bin"), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-w", "--offload-arch=gfx950", "-o", os.path.join(HERE, "bin", "ubench_fetch"), "/tmp/ubench_fetch.hip"], check=True)


if __name__ == "__main__":
    main()
