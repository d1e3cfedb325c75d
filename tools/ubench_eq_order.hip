#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#include "/tmp/eq_v0.inc"
__global__ void k_v0(float* buf, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    const int lane = threadIdx.x;
    float s1 = 0.f, s2 = 0.f, o = lane * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    float x[8], nx[8];
    for (int e = 0; e < 8; ++e) x[e] = 0.01f * (e + lane);
    float* sp = buf + 4096 + (lane >> 4) * 256 + (lane & 15);
    const float* lp = buf + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(BODY_v0
            : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c), [n0] "=&v"(nx[0]), [n1] "=&v"(nx[1]), [n2] "=&v"(nx[2]), [n3] "=&v"(nx[3]),
              [n4] "=&v"(nx[4]), [n5] "=&v"(nx[5]), [n6] "=&v"(nx[6]), [n7] "=&v"(nx[7])
            : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [sp] "v"(sp), [lp] "v"(lp), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]),
              [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]), [shr] "n"(6)
            : "v100","v102","v104","v106","v108","v109","v110","v111","v112","v113","v114","memory");
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]) :: "memory");
        for (int e = 0; e < 8; ++e) x[e] = nx[e];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) res[0] = t1 - t0;
    buf[8192 + lane] = s1 + s2 + o + c;
}
#include "/tmp/eq_v1.inc"
__global__ void k_v1(float* buf, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    const int lane = threadIdx.x;
    float s1 = 0.f, s2 = 0.f, o = lane * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    float x[8], nx[8];
    for (int e = 0; e < 8; ++e) x[e] = 0.01f * (e + lane);
    float* sp = buf + 4096 + (lane >> 4) * 256 + (lane & 15);
    const float* lp = buf + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(BODY_v1
            : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c), [n0] "=&v"(nx[0]), [n1] "=&v"(nx[1]), [n2] "=&v"(nx[2]), [n3] "=&v"(nx[3]),
              [n4] "=&v"(nx[4]), [n5] "=&v"(nx[5]), [n6] "=&v"(nx[6]), [n7] "=&v"(nx[7])
            : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [sp] "v"(sp), [lp] "v"(lp), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]),
              [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]), [shr] "n"(6)
            : "v100","v102","v104","v106","v108","v109","v110","v111","v112","v113","v114","memory");
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]) :: "memory");
        for (int e = 0; e < 8; ++e) x[e] = nx[e];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) res[0] = t1 - t0;
    buf[8192 + lane] = s1 + s2 + o + c;
}
#include "/tmp/eq_v2.inc"
__global__ void k_v2(float* buf, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    const int lane = threadIdx.x;
    float s1 = 0.f, s2 = 0.f, o = lane * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    float x[8], nx[8];
    for (int e = 0; e < 8; ++e) x[e] = 0.01f * (e + lane);
    float* sp = buf + 4096 + (lane >> 4) * 256 + (lane & 15);
    const float* lp = buf + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(BODY_v2
            : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c), [n0] "=&v"(nx[0]), [n1] "=&v"(nx[1]), [n2] "=&v"(nx[2]), [n3] "=&v"(nx[3]),
              [n4] "=&v"(nx[4]), [n5] "=&v"(nx[5]), [n6] "=&v"(nx[6]), [n7] "=&v"(nx[7])
            : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [sp] "v"(sp), [lp] "v"(lp), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]),
              [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]), [shr] "n"(6)
            : "v100","v102","v104","v106","v108","v109","v110","v111","v112","v113","v114","memory");
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]) :: "memory");
        for (int e = 0; e < 8; ++e) x[e] = nx[e];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) res[0] = t1 - t0;
    buf[8192 + lane] = s1 + s2 + o + c;
}
#include "/tmp/eq_v3.inc"
__global__ void k_v3(float* buf, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    const int lane = threadIdx.x;
    float s1 = 0.f, s2 = 0.f, o = lane * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    float x[8], nx[8];
    for (int e = 0; e < 8; ++e) x[e] = 0.01f * (e + lane);
    float* sp = buf + 4096 + (lane >> 4) * 256 + (lane & 15);
    const float* lp = buf + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(BODY_v3
            : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c), [n0] "=&v"(nx[0]), [n1] "=&v"(nx[1]), [n2] "=&v"(nx[2]), [n3] "=&v"(nx[3]),
              [n4] "=&v"(nx[4]), [n5] "=&v"(nx[5]), [n6] "=&v"(nx[6]), [n7] "=&v"(nx[7])
            : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [sp] "v"(sp), [lp] "v"(lp), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]),
              [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]), [shr] "n"(6)
            : "v100","v102","v104","v106","v108","v109","v110","v111","v112","v113","v114","memory");
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]) :: "memory");
        for (int e = 0; e < 8; ++e) x[e] = nx[e];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) res[0] = t1 - t0;
    buf[8192 + lane] = s1 + s2 + o + c;
}
#include "/tmp/eq_v4.inc"
__global__ void k_v4(float* buf, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    const int lane = threadIdx.x;
    float s1 = 0.f, s2 = 0.f, o = lane * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    float x[8], nx[8];
    for (int e = 0; e < 8; ++e) x[e] = 0.01f * (e + lane);
    float* sp = buf + 4096 + (lane >> 4) * 256 + (lane & 15);
    const float* lp = buf + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(BODY_v4
            : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c), [n0] "=&v"(nx[0]), [n1] "=&v"(nx[1]), [n2] "=&v"(nx[2]), [n3] "=&v"(nx[3]),
              [n4] "=&v"(nx[4]), [n5] "=&v"(nx[5]), [n6] "=&v"(nx[6]), [n7] "=&v"(nx[7])
            : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [sp] "v"(sp), [lp] "v"(lp), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]),
              [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]), [shr] "n"(6)
            : "v100","v102","v104","v106","v108","v109","v110","v111","v112","v113","v114","memory");
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]) :: "memory");
        for (int e = 0; e < 8; ++e) x[e] = nx[e];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) res[0] = t1 - t0;
    buf[8192 + lane] = s1 + s2 + o + c;
}
#include "/tmp/eq_v5.inc"
__global__ void k_v5(float* buf, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    const int lane = threadIdx.x;
    float s1 = 0.f, s2 = 0.f, o = lane * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    float x[8], nx[8];
    for (int e = 0; e < 8; ++e) x[e] = 0.01f * (e + lane);
    float* sp = buf + 4096 + (lane >> 4) * 256 + (lane & 15);
    const float* lp = buf + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(BODY_v5
            : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c), [n0] "=&v"(nx[0]), [n1] "=&v"(nx[1]), [n2] "=&v"(nx[2]), [n3] "=&v"(nx[3]),
              [n4] "=&v"(nx[4]), [n5] "=&v"(nx[5]), [n6] "=&v"(nx[6]), [n7] "=&v"(nx[7])
            : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [sp] "v"(sp), [lp] "v"(lp), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]),
              [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]), [shr] "n"(6)
            : "v100","v102","v104","v106","v108","v109","v110","v111","v112","v113","v114","memory");
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]) :: "memory");
        for (int e = 0; e < 8; ++e) x[e] = nx[e];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) res[0] = t1 - t0;
    buf[8192 + lane] = s1 + s2 + o + c;
}
#include "/tmp/eq_v6.inc"
__global__ void k_v6(float* buf, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    const int lane = threadIdx.x;
    float s1 = 0.f, s2 = 0.f, o = lane * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    float x[8], nx[8];
    for (int e = 0; e < 8; ++e) x[e] = 0.01f * (e + lane);
    float* sp = buf + 4096 + (lane >> 4) * 256 + (lane & 15);
    const float* lp = buf + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(BODY_v6
            : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c), [n0] "=&v"(nx[0]), [n1] "=&v"(nx[1]), [n2] "=&v"(nx[2]), [n3] "=&v"(nx[3]),
              [n4] "=&v"(nx[4]), [n5] "=&v"(nx[5]), [n6] "=&v"(nx[6]), [n7] "=&v"(nx[7])
            : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [sp] "v"(sp), [lp] "v"(lp), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]),
              [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]), [shr] "n"(6)
            : "v100","v102","v104","v106","v108","v109","v110","v111","v112","v113","v114","memory");
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]) :: "memory");
        for (int e = 0; e < 8; ++e) x[e] = nx[e];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) res[0] = t1 - t0;
    buf[8192 + lane] = s1 + s2 + o + c;
}
#include "/tmp/eq_v7.inc"
__global__ void k_v7(float* buf, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    const int lane = threadIdx.x;
    float s1 = 0.f, s2 = 0.f, o = lane * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    float x[8], nx[8];
    for (int e = 0; e < 8; ++e) x[e] = 0.01f * (e + lane);
    float* sp = buf + 4096 + (lane >> 4) * 256 + (lane & 15);
    const float* lp = buf + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(BODY_v7
            : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c), [n0] "=&v"(nx[0]), [n1] "=&v"(nx[1]), [n2] "=&v"(nx[2]), [n3] "=&v"(nx[3]),
              [n4] "=&v"(nx[4]), [n5] "=&v"(nx[5]), [n6] "=&v"(nx[6]), [n7] "=&v"(nx[7])
            : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [sp] "v"(sp), [lp] "v"(lp), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]),
              [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]), [shr] "n"(6)
            : "v100","v102","v104","v106","v108","v109","v110","v111","v112","v113","v114","memory");
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]) :: "memory");
        for (int e = 0; e < 8; ++e) x[e] = nx[e];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) res[0] = t1 - t0;
    buf[8192 + lane] = s1 + s2 + o + c;
}

typedef void (*KF)(float*, unsigned long long*, int, float, float, float, float, float);
int main() {
    float* buf; unsigned long long* res; hipMalloc(&buf, 65536); hipMemset(buf, 0, 65536); hipMalloc(&res, 8);
    const int iters = 4000;
    KF ks[] = {k_v0,k_v1,k_v2,k_v3,k_v4,k_v5,k_v6,k_v7};
    const char* nm[] = {"O T NOP Y A C D N M P","O T A Y C D N M P","O A T Y C D N M P","O T A Y N C D M P","O A T Y N C D M P","O A T Y C N D M P","O T A Y C N D M P","O A T Y C D M N P"};
    for (int k = 0; k < 8; ++k) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(ks[k], dim3(1), dim3(64), 0, 0, buf, res, iters, 0.9f, -0.4f, 0.3f, -0.5f, 0.2f); hipDeviceSynchronize(); }
        unsigned long long h; hipMemcpy(&h, res, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.2f cycles per step\n", nm[k], (double)h / iters / 128.0);
    }
    return 0;
}
