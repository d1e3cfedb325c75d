#!/usr/bin/env python3
"""Cross-lane data movement on gfx950: per-wave s_memtime ticks per instruction for the candidates of an
LDS-free register<->lane transpose (wave FFT exchanges) and of a wider EQ systolic row, 1..16 waves per
workgroup (16 = 4 per SIMD).  Same harness as ubench_fetch.py.
Writes + compiles tools/bin/ubench_xlane;   gpurun -- tools/bin/ubench_xlane"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
FULL = "row_mask:0xf bank_mask:0xf"
N = 512
CASES = {
    "v_add_f32 e32 (ref)": ["v_add_f32 v{d}, v2, v3"],
    "v_permlane16_swap": ["v_permlane16_swap_b32 v{d}, v{e}"],
    "v_permlane32_swap": ["v_permlane32_swap_b32 v{d}, v{e}"],
    "permlane32_swap dependent": ["v_permlane32_swap_b32 v20, v21"],
    "v_cndmask_dpp quad_perm": ["v_cndmask_b32_dpp v{d}, v2, v3, vcc quad_perm:[1,0,3,2] " + FULL],
    "v_cndmask_dpp row_shr:4": ["v_cndmask_b32_dpp v{d}, v2, v3, vcc row_shr:4 " + FULL],
    "v_mov_dpp row_shr:4 bank": ["v_mov_b32_dpp v{d}, v2 row_shr:4 row_mask:0xf bank_mask:0xa"],
    "v_cndmask e32 (no dpp)": ["v_cndmask_b32 v{d}, v2, v3, vcc"],
    "v_add_dpp wave_ror:1": ["v_add_f32_dpp v{d}, v2, v3 wave_ror:1 " + FULL],
    "v_add_dpp wave_shr:1": ["v_add_f32_dpp v{d}, v2, v3 wave_shr:1 " + FULL],
    "v_add_dpp row_ror:1": ["v_add_f32_dpp v{d}, v2, v3 row_ror:1 " + FULL],
    "dep add_dpp wave_ror (+nop1)": ["v_add_f32_dpp v20, v20, v3 wave_ror:1 " + FULL, "s_nop 1"],
    "dep add_dpp row_ror (+nop1)": ["v_add_f32_dpp v20, v20, v3 row_ror:1 " + FULL, "s_nop 1"],
    "v_pk_fma_f32": ["v_pk_fma_f32 v[{d}:{e}], v[2:3], v[4:5], v[2:3]"],
    "v_fma_f32": ["v_fma_f32 v{d}, v2, v3, v4"],
    "v_mul_f32 literal": ["v_mul_f32 v{d}, 0x3f6c835e, v3"],
    "v_mul_f32 sgpr operand": ["v_mul_f32 v{d}, s20, v3"],
    "v_fma_f32 sgpr operand": ["v_fma_f32 v{d}, v2, s20, v3"],
    "v_fmamk_f32 literal": ["v_fmamk_f32 v{d}, v2, 0x3f3504f3, v3"],
    "v_cndmask e64 sgpr mask": ["v_cndmask_b32_e64 v{d}, v2, v3, s[22:23]"],
    "v_cmp_lt_f32 (writes vcc)": ["v_cmp_lt_f32 vcc, v2, v3"],
    "v_add_f32 e64 (8B)": ["v_add_f32_e64 v{d}, v2, v3"],
    "v_mov_b32": ["v_mov_b32 v{d}, v2"],
    "v_lshl_add_u64": ["v_lshl_add_u64 v[{d}:{e}], v[2:3], 2, v[4:5]"],
    "ds_bpermute_b32": ["ds_bpermute_b32 v{d}, v6, v2"],
    "ds_swizzle": ["ds_swizzle_b32 v{d}, v2 offset:swizzle(SWAP,16)"],
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
            if "ds_" in pat[0] and k % 8 == 0:
                lines.append("s_waitcnt lgkmcnt(0)")
        body = "\\n".join(lines) + "\\n"
        clob = ", ".join(f'"v{r}"' for r in range(20, 36))
        src.append(f"""__global__ void k{i}(unsigned long long* res, int iters)
{{
    asm volatile("v_mov_b32 v2, 1.0\\nv_mov_b32 v3, 0.5\\nv_mov_b32 v4, 0.5\\nv_mov_b32 v5, 0.25\\nv_mov_b32 v20, 0\\nv_mov_b32 v21, 0\\nv_mbcnt_lo_u32_b32 v6, -1, 0\\nv_mbcnt_hi_u32_b32 v6, -1, v6\\nv_xor_b32 v6, 63, v6\\nv_lshlrev_b32 v6, 2, v6\\ns_mov_b32 vcc_lo, 0x55555555\\ns_mov_b32 vcc_hi, 0x55555555\\ns_mov_b32 s20, 0x3f000000\\ns_mov_b32 s22, 0x55555555\\ns_mov_b32 s23, 0x55555555" ::: "v2","v3","v4","v5","v6","v20","v21","vcc","s20","s22","s23");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
        asm volatile(".p2align 6\\n{body}" ::: "v2","v3","v4","v5",{clob},"memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) res[threadIdx.x >> 6] = t1 - t0;
}}""")
    src.append("int main() {\n    unsigned long long* res; hipMalloc(&res, 1024);\n    const int iters = 200; unsigned long long h[16];")
    src.append('    printf("%-30s", "waves per workgroup:"); for (int w : {1, 2, 4, 8, 16}) printf(" %7d", w); printf("\\n");')
    for i, n in enumerate(names):
        per = len(CASES[n])
        cnt = N + (per - N % per) % per
        src.append(f'    printf("%-30s", "{n}");\n    for (int w : {{1, 2, 4, 8, 16}}) {{\n'
                   f'        for (int r = 0; r < 2; ++r) {{ hipLaunchKernelGGL(k{i}, dim3(1), dim3(64 * w), 0, 0, res, iters); hipDeviceSynchronize(); }}\n'
                   f'        hipMemcpy(h, res, 8 * w, hipMemcpyDeviceToHost); unsigned long long m = 0; for (int q = 0; q < w; ++q) m = h[q] > m ? h[q] : m;\n'
                   f'        printf(" %7.2f", (double)m / (iters * {cnt}.0)); }}\n    printf("\\n"); fflush(stdout);')
    src.append("    return 0;\n}")
    open("/tmp/ubench_xlane.hip", "w").write("\n".join(src))
    os.makedirs(os.path.join(HERE, "bin"), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-w", "--offload-arch=gfx950", "-o", os.path.join(HERE, "bin", "ubench_xlane"), "/tmp/ubench_xlane.hip"], check=True)


if __name__ == "__main__":
    main()
