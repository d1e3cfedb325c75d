#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define EQ_NOP "s_nop 0\n"
#define EQ_YS(RY) "v_mov_b32_dpp " RY ", v100 row_shr:%[shr] row_mask:0xf bank_mask:0xf\n"
#define EQ_Y0(RY) "v_mov_b32 " RY ", v100\n"
#define EQ_C(RY, RC) "v_mov_b32_dpp " RY ", " RC " row_shl:1 row_mask:0xf bank_mask:0xf\n"
#define EQ_D(RC) "v_mov_b32_dpp " RC ", v100 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define EQ_M(RX) "v_mul_f32 v114, %[b0], " RX "\n"
#define EQ_P(RXP) "v_pk_mul_f32 v[108:109], %[b12], " RXP " op_sel_hi:[1,0]\n"
#define EQ_O "v_add_f32 v100, v112, v114\n"
#define EQ_T "v_add_f32 v108, v113, v108\n"
#define EQ_A "v_pk_mul_f32 v[110:111], %[a12], v[100:101] op_sel_hi:[1,0]\n"
#define EQ_N "v_pk_add_f32 v[112:113], v[108:109], v[110:111] neg_lo:[0,1] neg_hi:[0,1]\n"
#define EQ_BODY \
        EQ_Y("v104") EQ_C("v104", "v102") EQ_D("v102") EQ_M("v102") EQ_P("v[102:103]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v106") EQ_A EQ_C("v106", "v104") EQ_D("v104") EQ_N EQ_M("v104") EQ_P("v[104:105]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v102") EQ_A EQ_C("v102", "v106") EQ_D("v106") EQ_N EQ_M("v106") EQ_P("v[106:107]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v104") EQ_A EQ_C("v104", "v102") EQ_D("v102") EQ_N EQ_M("v102") EQ_P("v[102:103]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v106") EQ_A EQ_C("v106", "v104") EQ_D("v104") EQ_N EQ_M("v104") EQ_P("v[104:105]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v102") EQ_A EQ_C("v102", "v106") EQ_D("v106") EQ_N EQ_M("v106") EQ_P("v[106:107]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v104") EQ_A EQ_C("v104", "v102") EQ_D("v102") EQ_N EQ_M("v102") EQ_P("v[102:103]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v106") EQ_A EQ_C("v106", "v104") EQ_D("v104") EQ_N EQ_M("v104") EQ_P("v[104:105]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v102") EQ_A EQ_C("v102", "v106") EQ_D("v106") EQ_N EQ_M("v106") EQ_P("v[106:107]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v104") EQ_A EQ_C("v104", "v102") EQ_D("v102") EQ_N EQ_M("v102") EQ_P("v[102:103]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v106") EQ_A EQ_C("v106", "v104") EQ_D("v104") EQ_N EQ_M("v104") EQ_P("v[104:105]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v102") EQ_A EQ_C("v102", "v106") EQ_D("v106") EQ_N EQ_M("v106") EQ_P("v[106:107]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v104") EQ_A EQ_C("v104", "v102") EQ_D("v102") EQ_N EQ_M("v102") EQ_P("v[102:103]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v106") EQ_A EQ_C("v106", "v104") EQ_D("v104") EQ_N EQ_M("v104") EQ_P("v[104:105]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v102") EQ_A EQ_C("v102", "v106") EQ_D("v106") EQ_N EQ_M("v106") EQ_P("v[106:107]") \
        EQ_O EQ_T EQ_NOP EQ_Y("v104") EQ_A EQ_C("v104", "v102") EQ_D("v102") EQ_N EQ_M("v102") EQ_P("v[102:103]") \
        EQ_O EQ_T EQ_A EQ_N

#define EQ_IN "v_mov_b32 v100, %[o]\n v_mov_b32 v112, %[s1]\n v_mov_b32 v113, %[s2]\n v_mov_b32 v102, %[c]\n"
#define EQ_OUT "v_mov_b32 %[s1], v112\n v_mov_b32 %[s2], v113\n v_mov_b32 %[o], v100\n v_mov_b32 %[c], v104\n"
#define EQ_CLOBBERS "v100", "v102", "v104", "v106", "v108", "v109", "v110", "v111", "v112", "v113", "v114"

#define EQ_Y(RY) EQ_YS(RY)

__global__ void k0(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k1(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k2(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k3(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k4(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k5(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k6(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k7(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k8(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k9(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k10(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k11(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k12(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k13(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k14(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

__global__ void k15(float* out, unsigned long long* res, int iters, float b0, float b1, float b2, float a1, float a2)
{
    float s1 = 0.f, s2 = 0.f, o = threadIdx.x * 1e-3f, c = 0.5f;
    v2f b12 = {b1, b2}, a12 = {a1, a2};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(".p2align 6\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"  EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                     : EQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    out[threadIdx.x] = s1 + s2 + o + c;
}

typedef void (*KF)(float*, unsigned long long*, int, float, float, float, float, float);
int main() {
    float* out; unsigned long long* res; hipMalloc(&out, 256); hipMalloc(&res, 8);
    const int iters = 20000;
    KF ks[16] = {k0,k1,k2,k3,k4,k5,k6,k7,k8,k9,k10,k11,k12,k13,k14,k15};
    for (int k = 0; k < 16; ++k) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(ks[k], dim3(1), dim3(64), 0, 0, out, res, iters, 0.9f, -0.4f, 0.3f, -0.5f, 0.2f); hipDeviceSynchronize(); }
        unsigned long long h; hipMemcpy(&h, res, 8, hipMemcpyDeviceToHost);
        printf("start offset %2d bytes mod 64: %.2f cycles per step\n", 4 * k, (double)h / iters / 16.0);
    }
    return 0;
}
