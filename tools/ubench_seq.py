#!/usr/bin/env python3
"""Cycles per repetition of short instruction sequences on a lone gfx950 wave (64 repetitions unrolled,
64-byte aligned).  Edit SEQS, run to write + compile tools/bin/ubench_seq;  gpurun -- tools/bin/ubench_seq"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
F = "row_mask:0xf bank_mask:0xf"
P = "v_pk_mul_f32 v[106:107], v[4:5], v[102:103] op_sel_hi:[1,0]"
M = "v_mul_f32_dpp v109, v102, v6 row_ror:1 " + F
O = "v_add_f32_dpp v102, v106, v112 row_ror:1 " + F
T = "v_add_f32_dpp v108, v107, v113 row_ror:1 " + F
A = "v_pk_mul_f32 v[110:111], v[8:9], v[102:103] op_sel_hi:[1,0]"
N = "v_pk_add_f32 v[112:113], v[108:109], v[110:111] neg_lo:[0,1] neg_hi:[0,1]"
STEP = [O, T, A, P, N, M]
CAP = "v_mov_b32_dpp v100, v102 row_ror:8 row_mask:0xf bank_mask:0x2"
INJ = "v_mov_b32_dpp v102, v14 row_ror:8 row_mask:0xf bank_mask:0x8"
CAPF = "v_mov_b32_dpp v100, v102 row_ror:8 " + F
INJF = "v_mov_b32_dpp v101, v14 row_ror:8 " + F
ST = "global_store_dword v[16:17], v100, off"
LD = "global_load_dword v115, v[16:17], off"
ST3 = "global_store_dword v[18:19], v100, off"          # lane ^ 3 address order
STS = "global_store_dword v20, v100, s[2:3]"            # saddr form, natural order
STX2 = "global_store_dwordx2 v[16:17], v[100:101], off"
STX4 = "global_store_dwordx4 v[16:17], v[100:103], off"
STEP = [O, T, A, P, N, M]
CAP = "v_mov_b32_dpp v100, v102 row_ror:8 row_mask:0xf bank_mask:0x2"
INJ = "v_mov_b32_dpp v102, v14 row_ror:8 row_mask:0xf bank_mask:0x8"
ST = "global_store_dword v[16:17], v100, off"
STS = "global_store_dword v20, v100, s[2:3]"
STB = "buffer_store_dword v100, v20, s[4:7], 0 offen"
LD = "global_load_dword v115, v[16:17], off"
LDS_ = "global_load_dword v115, v20, s[2:3]"
PORTSTEP = [O, T, A, CAP, INJ, P, N, M]
SEQS = {
    "16 steps": STEP * 16,
    "16 steps + 4 ports": (STEP * 3 + PORTSTEP) * 4,
    "16 steps + store (64-bit vaddr)": STEP * 15 + [O, T, A, ST, P, N, M],
    "16 steps + store (saddr + 32-bit voffset)": STEP * 15 + [O, T, A, STS, P, N, M],
    "16 steps + buffer_store offen": STEP * 15 + [O, T, A, STB, P, N, M],
    "16 steps + load (64-bit vaddr)": STEP * 15 + [O, T, A, LD, P, N, M],
    "16 steps + load (saddr)": STEP * 15 + [O, T, A, LDS_, P, N, M],
    "16 steps + 4 ports + store + load": (STEP * 3 + PORTSTEP) * 3 + STEP * 3 + [O, T, A, CAP, ST, INJ, LD, P, N, M],
    "16 steps + 4 ports + store + load (saddr)": (STEP * 3 + PORTSTEP) * 3 + STEP * 3 + [O, T, A, CAP, STS, INJ, LDS_, P, N, M],
}


def main():
    src = ["#include <hip/hip_runtime.h>", "#include <cstdio>"]
    names = list(SEQS)
    clob = ", ".join(f'"v{r}"' for r in list(range(100, 128)))
    for i, n in enumerate(names):
        body = "\\n".join(SEQS[n] * 1) + "\\n"
        src.append(f"""__global__ void k{i}(unsigned long long* res, int iters, float* buf)
{{
    asm volatile("v_mov_b32 v16, %0\\nv_mov_b32 v17, %1" :: "v"((unsigned)(size_t)(buf + threadIdx.x)), "v"((unsigned)((size_t)(buf + threadIdx.x) >> 32)) : "v16", "v17");
    asm volatile("v_mov_b32 v18, %0\\nv_mov_b32 v19, %1\\nv_mov_b32 v20, %2\\ns_mov_b64 s[2:3], %3" :: "v"((unsigned)(size_t)(buf + (threadIdx.x ^ 3))), "v"((unsigned)((size_t)(buf + threadIdx.x) >> 32)), "v"(threadIdx.x * 4), "s"(buf) : "v18", "v19", "v20", "s2", "s3");
    asm volatile("s_mov_b32 s4, s2\\ns_mov_b32 s5, s3\\ns_mov_b32 s6, 0x10000\\ns_mov_b32 s7, 0x00020000" ::: "s4", "s5", "s6", "s7");
    asm volatile("v_mov_b32 v4, 0.5\\nv_mov_b32 v5, 0.25\\nv_mov_b32 v6, 0.5\\nv_mov_b32 v8, 0.5\\nv_mov_b32 v9, 0.25\\nv_mov_b32 v10, 0.5\\nv_mov_b32 v11, 0.25\\nv_mov_b32 v12, 0.5\\nv_mov_b32 v13, 0.25\\n"
                 "v_mov_b32 v7, 0.5\\nv_mov_b32 v14, 0.5\\nv_mov_b32 v15, 0.5\\nv_mov_b32 v102, 0\\nv_mov_b32 v103, 0\\nv_mov_b32 v104, 0\\nv_mov_b32 v105, 0\\nv_mov_b32 v106, 0\\nv_mov_b32 v107, 0\\nv_mov_b32 v108, 0\\nv_mov_b32 v109, 0\\nv_mov_b32 v112, 0\\nv_mov_b32 v113, 0\\nv_mov_b32 v114, 0"
                 ::: "v4","v5","v6","v7","v15","v8","v9","v10","v11","v12","v13",{clob});
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
        asm volatile(".p2align 6\\n{body}" ::: {clob},"v14","v16","v17","v18","v19","v20","s2","s3","s4","s5","s6","s7","v115","memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
}}""")
    src.append("int main() {\n    unsigned long long* res; hipMalloc(&res, 1024); float* buf; hipMalloc(&buf, 65536);\n    const int iters = 300; unsigned long long h;")
    for i, n in enumerate(names):
        src.append(f'    for (int r = 0; r < 2; ++r) {{ hipLaunchKernelGGL(k{i}, dim3(1), dim3(64), 0, 0, res, iters, buf); hipDeviceSynchronize(); }}\n'
                   f'    hipMemcpy(&h, res, 8, hipMemcpyDeviceToHost); printf("%-40s %6.2f cycles / repetition (%d instructions, %.2f per sample)\\n", "{n}", (double)h / (iters * 1.0), {len(SEQS[n])}, (double)h / (iters * 16.0));')
    src.append('    return 0;\n}')
    open("/tmp/ubench_seq.hip", "w").write("\n".join(src))
    os.makedirs(os.path.join(HERE, "bin"), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-w", "--offload-arch=gfx950", "-o", os.path.join(HERE, "bin", "ubench_seq"), "/tmp/ubench_seq.hip"], check=True)


if __name__ == "__main__":
    main()
