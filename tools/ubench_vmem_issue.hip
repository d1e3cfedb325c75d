// What a lone wave pays per vector-memory instruction between 4-cycle VALU instructions (the k_eq_ring situation:
// one store + one load per 16 steps of 6.5 packed / DPP instructions), by addressing form:
//   0 none   1 global_* with SGPR base + VGPR offset (what k_eq_ring uses)   2 global_* with 64-bit VGPR addresses
//   3 buffer_* (MUBUF, offen)   4 store only (global saddr)   5 load only (global saddr)
// hipcc --offload-arch=gfx950 -O3 tools/ubench_vmem_issue.hip -o tools/bin/ubench_vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));

#define VALU8 "v_pk_mul_f32 %[a], %[a], %[c]\n v_pk_add_f32 %[b], %[b], %[a]\n v_pk_mul_f32 %[a], %[a], %[c]\n v_pk_add_f32 %[b], %[b], %[a]\n" \
              "v_pk_mul_f32 %[a], %[a], %[c]\n v_pk_add_f32 %[b], %[b], %[a]\n v_pk_mul_f32 %[a], %[a], %[c]\n v_pk_add_f32 %[b], %[b], %[a]\n"
#define VALU104 VALU8 VALU8 VALU8 VALU8 VALU8 VALU8 VALU8 VALU8 VALU8 VALU8 VALU8 VALU8 VALU8

template <int MODE>
__global__ void k(float *buf, unsigned long long *res, int iters)
{
    const unsigned lane = threadIdx.x;
    v2f a = {1.0f + lane * 1e-6f, 1.0f}, b = {0.f, 0.f}, c = {0.999999f, 1.000001f};
    unsigned off = lane * 4;
    float ld = 0.f, st = (float)lane;
    float *p64 = buf + lane;
    i4 rsrc;
    {
        const unsigned long long base = (unsigned long long)buf;
        rsrc.x = (int)(unsigned)base; rsrc.y = (int)(unsigned)(base >> 32); rsrc.z = 1 << 20; rsrc.w = 0x00020000;
        rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x); rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y);
        rsrc.z = __builtin_amdgcn_readfirstlane(rsrc.z); rsrc.w = __builtin_amdgcn_readfirstlane(rsrc.w);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        // fresh lines every iteration (repeated accesses to one line serialise in the L1): a 512 KiB window
        off = (off + 256u) & 0x7ffffu;
        p64 = buf + (off >> 2);
        if (MODE == 0)
            asm volatile(VALU104 : [a] "+v"(a), [b] "+v"(b) : [c] "v"(c));
        else if (MODE == 1)
            asm volatile(VALU104 "global_store_dword %[off], %[st], %[base] offset:1024\n global_load_dword %[ld], %[off], %[base]\n"
                         : [a] "+v"(a), [b] "+v"(b), [ld] "=&v"(ld) : [c] "v"(c), [off] "v"(off), [st] "v"(st), [base] "s"(buf) : "memory");
        else if (MODE == 2)
            asm volatile(VALU104 "global_store_dword %[p], %[st], off offset:1024\n global_load_dword %[ld], %[p], off\n"
                         : [a] "+v"(a), [b] "+v"(b), [ld] "=&v"(ld) : [c] "v"(c), [p] "v"(p64), [st] "v"(st) : "memory");
        else if (MODE == 3)
            asm volatile(VALU104 "buffer_store_dword %[st], %[off], %[r], 0 offen offset:1024\n buffer_load_dword %[ld], %[off], %[r], 0 offen\n"
                         : [a] "+v"(a), [b] "+v"(b), [ld] "=&v"(ld) : [c] "v"(c), [off] "v"(off), [st] "v"(st), [r] "s"(rsrc) : "memory");
        else if (MODE == 4)
            asm volatile(VALU104 "global_store_dword %[off], %[st], %[base] offset:1024\n"
                         : [a] "+v"(a), [b] "+v"(b) : [c] "v"(c), [off] "v"(off), [st] "v"(st), [base] "s"(buf) : "memory");
        else
            asm volatile(VALU104 "global_load_dword %[ld], %[off], %[base]\n"
                         : [a] "+v"(a), [b] "+v"(b), [ld] "=&v"(ld) : [c] "v"(c), [off] "v"(off), [base] "s"(buf) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) res[0] = t1 - t0;
    buf[(600 << 10 >> 2) + lane] = a.x + a.y + b.x + b.y + ld + (float)off;
}

template <int MODE>
static double run(float *buf, unsigned long long *res, int iters)
{
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, buf, res, iters);
        hipDeviceSynchronize();
    }
    unsigned long long h = 0;
    hipMemcpy(&h, res, 8, hipMemcpyDeviceToHost);
    return (double)h / iters;
}

int main()
{
    float *buf; unsigned long long *res;
    hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20); hipMalloc(&res, 8);
    const int iters = 20000;
    const double base = run<0>(buf, res, iters);
    printf("104 packed VALU instructions: %.1f cycles (%.2f each)\n", base, base / 104);
    const char *names[] = {"", "global saddr store + load", "global 64-bit-address store + load", "buffer (offen) store + load",
                           "global saddr store only", "global saddr load only"};
    const double r[] = {0, run<1>(buf, res, iters), run<2>(buf, res, iters), run<3>(buf, res, iters), run<4>(buf, res, iters),
                        run<5>(buf, res, iters)};
    for (int m = 1; m <= 5; ++m) printf("+ %-36s: %.1f cycles (+%.1f)\n", names[m], r[m], r[m] - base);
    return 0;
}
