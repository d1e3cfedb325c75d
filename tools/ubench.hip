// tools/ubench.hip -- single-wave instruction-latency microbenchmarks for gfx950 (tuning aid only).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o gpurun_out/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))

struct Res { unsigned long long cyc, rt; };

template <int MODE>
__global__ void k(float *out, Res *res, float c, int iters)
{
    float a = threadIdx.x * 0.001f, b = a + 1.f, d = a + 2.f, e = a + 3.f;
    float f = a + 4.f, g = a + 5.f, h = a + 6.f, i2 = a + 7.f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p = {a, b}, q = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP256(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));) }
        if (MODE == 1) { REP256(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                             : "+v"(a), "+v"(b), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i2) : "v"(c));) }
        if (MODE == 2) { REP256(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));) }
        if (MODE == 3) { REP256(asm volatile("s_nop 1\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32 %1, %1, %2\n v_add_f32 %0, %1, %2" : "+v"(a), "+v"(b) : "v"(c));) }
        if (MODE == 4) { REP256(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));) }
        if (MODE == 5) { REP256(asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(c));) }
        if (MODE == 6) { REP256(asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c));) }
        if (MODE == 7) { REP256(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));) }
        if (MODE == 8) { REP256(asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(c));) }
        if (MODE == 9) { REP256(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(0));) }
        if (MODE == 10) { REP256(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(d), "+v"(e) : "v"(c));) }
        if (MODE == 11) { REP256(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));) }
        if (MODE == 12) { REP256(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(c));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { res->cyc = t1 - t0; res->rt = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + d + e + f + g + h + i2 + p.x + p.y;
}

template <int MODE>
void run(const char *name, int ops_per_rep, int blocks)
{
    float *out; Res *res;
    hipMalloc(&out, blocks * 64 * sizeof(float)); hipMalloc(&res, sizeof(Res));
    const int iters = 200;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, res, 1.0001f, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, res, 1.0001f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    Res h; hipMemcpy(&h, res, sizeof(h), hipMemcpyDeviceToHost);
    const double n = 256.0 * iters * ops_per_rep;
    printf("%-34s blocks=%4d  %.2f cyc/instr  clock %.0f MHz  wall %.1f ns/instr\n", name, blocks,
           h.cyc / n, (double)h.cyc / (double)h.rt * 100.0, ms * 1e6 / n);
    hipFree(out); hipFree(res);
}

int main()
{
    for (int blocks : {1, 128, 1024}) {
        run<0>("dep v_add_f32", 1, blocks);
        run<8>("2 indep v_add chains", 2, blocks);
        run<10>("4 indep v_add chains", 4, blocks);
        run<1>("8 indep v_add chains", 8, blocks);
        run<7>("dep v_fma_f32", 1, blocks);
        run<2>("dep dpp mov (+s_nop 1)", 1, blocks);
        run<6>("dep v_add_f32_dpp", 1, blocks);
        run<3>("dep dpp->mul->add (3 instr)", 3, blocks);
        run<5>("dep mul->add->sub", 3, blocks);
        run<4>("dep v_pk_mul_f32", 1, blocks);
        run<11>("dep v_pk_add_f32", 1, blocks);
        run<12>("dep v_cndmask", 1, blocks);
        run<9>("dep ds_bpermute + wait", 1, blocks);
    }
    return 0;
}
