#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
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
#include "../open_headstage_amd/csrc/eq_group_asm.inc"
#define EQG_CLOBBERS "v100", "v102", "v104", "v106", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "memory"
__global__ void kcmp(const float* xin, float* outA, float* outB, float* stA, float* stB, float b0, float b1, float b2, float a1, float a2)
{
    const int lane = threadIdx.x;
    v2f b12 = {b1 + lane * 0.001f, b2}, a12 = {a1, a2 - lane * 0.002f};
    float bb0 = b0 + 0.01f * lane;
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = xin[e * 64 + lane];
    {   // A: old per-chunk asm
        float s1 = 0.1f, s2 = -0.2f, o = 0.3f * lane, c = 0.5f + lane;
        for (int e = 0; e < 8; ++e) {
            outA[e * 64 + lane] = c;
            c = x[e];
            asm volatile(EQ_IN EQ_BODY EQ_OUT
                         : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                         : [b0] "v"(bb0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(6)
                         : EQ_CLOBBERS);
        }
        stA[lane * 4 + 0] = s1; stA[lane * 4 + 1] = s2; stA[lane * 4 + 2] = o; stA[lane * 4 + 3] = c;
    }
    {   // B: new group asm; sp such that chunk e lands at outB[e*64 + lane] (64-byte steps => 16 floats per chunk!)
        float s1 = 0.1f, s2 = -0.2f, o = 0.3f * lane, c = 0.5f + lane;
        float* sp = outB + (lane >> 4) * 128 + (lane & 15);
        float nx[8];
        asm volatile("global_load_dword %0, %8, off\n global_load_dword %1, %8, off offset:64\n global_load_dword %2, %8, off offset:128\n"
                     "global_load_dword %3, %8, off offset:192\n global_load_dword %4, %8, off offset:256\n global_load_dword %5, %8, off offset:320\n"
                     "global_load_dword %6, %8, off offset:384\n global_load_dword %7, %8, off offset:448\n"
                     : "=&v"(nx[0]), "=&v"(nx[1]), "=&v"(nx[2]), "=&v"(nx[3]), "=&v"(nx[4]), "=&v"(nx[5]), "=&v"(nx[6]), "=&v"(nx[7])
                     : "v"(xin + lane) : "memory");
        asm volatile(EQ_GROUP_BODY_SHR
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(bb0), [b12] "v"(b12), [a12] "v"(a12), [sp] "v"(sp), [x0] "v"(x[0]), [x1] "v"(x[1]),
                       [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]), [shr] "n"(6)
                     : EQG_CLOBBERS);
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]) :: "memory");
        stB[lane * 4 + 0] = s1 + 0.f * (nx[0] + nx[7]); stB[lane * 4 + 1] = s2; stB[lane * 4 + 2] = o; stB[lane * 4 + 3] = c;
    }
}
int main() {
    float hx[512]; for (int i = 0; i < 512; ++i) hx[i] = sinf(0.37f * i) * 0.8f;
    float *dx, *oa, *ob, *sa, *sb;
    hipMalloc(&dx, 2048); hipMalloc(&oa, 4096); hipMalloc(&ob, 4096); hipMalloc(&sa, 1024); hipMalloc(&sb, 1024);
    hipMemcpy(dx, hx, 2048, hipMemcpyHostToDevice); hipMemset(oa, 0, 4096); hipMemset(ob, 0, 4096);
    hipLaunchKernelGGL(kcmp, dim3(1), dim3(64), 0, 0, dx, oa, ob, sa, sb, 0.9f, -0.4f, 0.3f, -0.5f, 0.2f);
    hipDeviceSynchronize();
    float ha[1024], hb[1024], ta[256], tb[256];
    hipMemcpy(ha, oa, 4096, hipMemcpyDeviceToHost); hipMemcpy(hb, ob, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(ta, sa, 1024, hipMemcpyDeviceToHost); hipMemcpy(tb, sb, 1024, hipMemcpyDeviceToHost);
    // A stores chunk e at [e*64 + lane]; B stores chunk e at byte offset 64e => float index 16e + lane
    int bad = 0;
    int sbad = 0;
    for (int e = 0; e < 8; ++e) for (int l = 0; l < 64; ++l) {
        const float a = ha[e * 64 + l], b = hb[(l >> 4) * 128 + 16 * e + (l & 15)];
        if (a != b) { if (sbad < 8) printf("store mismatch chunk %d lane %d: %g vs %g\n", e, l, a, b); ++sbad; }
    }
    printf("store mismatches: %d\n", sbad);
    for (int i = 0; i < 256; ++i) if (ta[i] != tb[i]) { if (bad < 10) printf("state mismatch lane %d field %d: %g vs %g\n", i / 4, i % 4, ta[i], tb[i]); ++bad; }
    printf("state mismatches: %d\n", bad);
    printf("A chunk1 lanes0-3: %g %g %g %g | B idx16-19: %g %g %g %g\n", ha[64], ha[65], ha[66], ha[67], hb[16], hb[17], hb[18], hb[19]);
    return 0;
}
