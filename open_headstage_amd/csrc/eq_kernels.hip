// eq_kernels.hip -- 10-band parametric EQ (DF2T biquad cascade) for gfx950.
//
// Computes exactly what src/dsp/parametric_eq.rs:116-122,166-179 computes
// (biquad 0.4.2 DirectForm2Transposed::run):
//     out = s1 + b0*x ;  s1 = (s2 + b1*x) - a1*out ;  s2 = b2*x - a2*out
// with every product and sum rounded separately -- this file is compiled with
// -ffp-contract=off and the result is BIT-EXACT against the CPU oracle.  An IIR
// is a serial recurrence in time, so the parallelism is (a) across chains
// (stream x channel) and (b) ACROSS BANDS, systolically:
//
//   a 16-lane DPP row is one chain; lane j of the row is band j and keeps that
//   band's (s1, s2, coefficients) in VGPRs for the whole launch.  Every step each
//   lane filters one sample and hands its output to lane j+1 with a single
//   `row_shr:1` DPP move, so band j works on sample t-j while band j+1 works on
//   sample t-j-1.  The arithmetic each band performs, and its order, are those
//   of the reference's sample-major / band-minor loop.
//
// Samples enter at lane 0 and leave at lane NB-1; a second register per lane (the "conveyor")
// carries 16 inputs in and 16 outputs out per chunk with one more DPP move per step, so each
// row loads / stores 64 contiguous bytes per 16 steps and the inner loop has no LDS or memory
// instruction at all.  More than 16 bands = several passes.
#include "kernels.h"

#include <cstdlib>

namespace ohs {

typedef float v2f __attribute__((ext_vector_type(2)));

// DPP move: lanes whose source lane lies outside their 16-lane row keep `old`.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL,
                                                      0xf, 0xf, false));
}
constexpr int kRowShr1 = 0x111;   // lane j <- lane j-1
constexpr int kRowShl1 = 0x101;   // lane j <- lane j+1

// One chunk = 16 systolic steps.  Register roles per lane j of a row:
//   prev : this band's output of the previous step (what lane j+1 filters next)
//   c    : the row's conveyor.  It enters a chunk holding 16 input samples (lane i = x[t0+i]),
//          moves one lane to the left per step so that lane 0 always sees the next input, and
//          takes one finished sample per step in at lane 15; it leaves the chunk holding 16
//          finished samples (lane i = y[t0 + i - NB]).
// SHR = 16 - NB brings the last band's (lane NB-1) output to lane 15; lanes >= NB are don't-care.
// SEL  : some band of this pass is disabled -> pass-through select (parametric_eq.rs:118-120).
// GATED: first / last chunks, where a lane's sample index t - j can fall outside [0, n).
template <int SHR, bool SEL, bool GATED>
__device__ __forceinline__ void eq_chunk(long long t0, long long n, int j, bool en,
                                         float b0, float b1, float b2, float a1, float a2,
                                         float &s1, float &s2, float &prev, float &c)
{
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float yv = prev;
        if constexpr (SHR != 0) yv = dpp_mov<0x110 + SHR>(prev, prev);
        const float x = dpp_mov<kRowShr1>(c, prev);    // lane 0 <- c (next input sample)
        c = dpp_mov<kRowShl1>(yv, c);                  // lane 15 <- finished sample
        const float out = s1 + b0 * x;
        const float n1 = (s2 + b1 * x) - a1 * out;
        const float n2 = b2 * x - a2 * out;
        if constexpr (GATED || SEL) {
            bool upd = en;
            if constexpr (GATED) {
                const long long u = t0 + k - j;        // sample index this lane is filtering
                upd = en && u >= 0 && u < n;
            }
            s1 = upd ? n1 : s1;
            s2 = upd ? n2 : s2;
            prev = upd ? out : x;
        } else {
            s1 = n1; s2 = n2; prev = out;
        }
    }
}

// Steady-state chunk (every lane's sample index valid, every band of the pass enabled): 16 steps
// in ONE hand-scheduled asm block.  A step is 9 VALU instructions in two 3-deep dependency chains
//     out(t) -> row_shr:1 -> b0*x -> s1 + .            (hand-over to the next band)
//     out(t) -> (a1,a2)*out -> (t2,b2x) - .  -> s1(t+1) (this band's recurrence)
// On gfx950 a lone wave issues a VALU op every ~4.5 cycles but a DEPENDENT op waits ~8 cycles
// (measured: tools/ubench.hip), so the chains are interleaved with the side work (conveyor moves,
// b1/b2 products): ~41 cycles per sample instead of the ~100 hipcc's own order reaches.
// The arithmetic is the reference's, operand for operand: 5 products and 4 sums, each rounded
// separately (v_pk_mul_f32 / v_pk_add_f32 are two independent IEEE f32 operations).
//   v100 = out        v[108:109] = (b1 x -> t2, b2 x)      v[112:113] = (s1, s2)
//   v114 = b0 x       v[110:111] = (a1 out, a2 out)
//   v102 / v104 / v106 rotate through the roles conveyor -> x -> scratch, so that the conveyor shift
//   itself provides the "old" lanes of the next DPP move and no register copy is needed.
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
template <int SHR>
__device__ __forceinline__ void eq_chunk_fast(float b0, v2f b12, v2f a12, float &s1, float &s2, float &o, float &c)
{
    if constexpr (SHR != 0) {
#define EQ_Y(RY) EQ_YS(RY)
        asm volatile(EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12), [shr] "n"(SHR)
                     : EQ_CLOBBERS);
#undef EQ_Y
    } else {
#define EQ_Y(RY) EQ_Y0(RY)
        asm volatile(EQ_IN EQ_BODY EQ_OUT
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [c] "+v"(c)
                     : [b0] "v"(b0), [b12] "v"(b12), [a12] "v"(a12)
                     : EQ_CLOBBERS);
#undef EQ_Y
    }
}

// Untracked (inline-asm) global accesses for the interior of the stream.  hipcc cannot count
// vmcnt across the chunk loop and falls back to `s_waitcnt vmcnt(0)` in front of every chunk,
// which serialises each chunk behind the store it has just issued (+~500 cycles per 16 samples).
// Here the queue is counted by hand: per group of G = 8 chunks one wave issues 8 prefetch loads,
// then 8 stores, and `s_waitcnt vmcnt(8)` (all but the 8 youngest = the stores) retires the loads.
__device__ __forceinline__ void eq_load8(float (&v)[8], const float *p)
{
    asm volatile("global_load_dword %0, %8, off\n"
                 "global_load_dword %1, %8, off offset:64\n"
                 "global_load_dword %2, %8, off offset:128\n"
                 "global_load_dword %3, %8, off offset:192\n"
                 "global_load_dword %4, %8, off offset:256\n"
                 "global_load_dword %5, %8, off offset:320\n"
                 "global_load_dword %6, %8, off offset:384\n"
                 "global_load_dword %7, %8, off offset:448\n"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]),
                   "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p)
                 : "memory");
}
template <int OFF>
__device__ __forceinline__ void eq_store1(float *p, float v)
{
    asm volatile("global_store_dword %0, %1, off offset:%2" : : "v"(p), "v"(v), "n"(OFF) : "memory");
}
// wait until at most N vector-memory operations of this wave are outstanding; ties the loaded
// registers to the wait so that no use of them can be scheduled above it
template <int N>
__device__ __forceinline__ void eq_wait_loads(float (&v)[8])
{
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "n"(N)
                 : "memory");
}

template <int SHR, bool SEL>
__global__ __launch_bounds__(64) void k_eq_pass(const float *in, float *out,
                                                long long stream_stride, long long ch_stride,
                                                long long n, int n_chains, const EqPassTable tab,
                                                float *__restrict__ state)
{
    constexpr int NB = 16 - SHR;
    constexpr int G = 8;            // chunks per group (128 samples per chain)
    const int lane = threadIdx.x;
    const int j = lane & 15;
    const long long chain = (long long)blockIdx.x * 4 + (lane >> 4);
    const bool live = chain < n_chains;
    const long long cc = live ? chain : 0;      // dead rows shadow chain 0 but never store
    const float *src = in + (cc >> 1) * stream_stride + (cc & 1) * ch_stride;
    float *dst = out + (cc >> 1) * stream_stride + (cc & 1) * ch_stride;

    const float b0 = tab.b0[j], b1 = tab.b1[j], b2 = tab.b2[j], a1 = tab.a1[j], a2 = tab.a2[j];
    const bool en = tab.en[j] != 0;
    float s1 = state[(cc * 16 + j) * 2 + 0];
    float s2 = state[(cc * 16 + j) * 2 + 1];
    float prev = 0.0f, c = 0.0f;

    // the finished sample y[i] enters the conveyor at step i + NB; chunk q covers steps 16q..16q+15
    const long long n_chunks = (n - 1 + NB) / 16 + 1;
    const long long n_groups = (n_chunks + G - 1) / G;
    // groups [1, g_fast_end) are "interior": all 8 chunks ungated, the prefetch of the next group and
    // every store in bounds, every row of the wave live
    long long g_fast_end = n / (16 * G) - 1;            // (g + 2) * 128 <= n
    const bool all_live = ((long long)blockIdx.x * 4 + 3) < n_chains;
    if (SEL || !all_live) g_fast_end = 0;

    float cur[G], nxt[G];
#pragma unroll
    for (int e = 0; e < G; ++e) {
        const long long xi = (long long)e * 16 + j;
        cur[e] = (live && xi < n) ? src[xi] : 0.0f;
    }
    for (long long g = 0; g < n_groups; ++g) {
        if (g >= 1 && g < g_fast_end) {
            if constexpr (!SEL) {
                const float *lp = src + (g + 1) * (16 * G) + j;
                float *sp = dst + g * (16 * G) - 16 - NB + j;
                eq_load8(nxt, lp);
#define OHS_EQ_FAST_CHUNK(E)                                                             \
                eq_store1<64 * E>(sp, c);                                                \
                c = cur[E];                                                              \
                eq_chunk_fast<SHR>(b0, (v2f){b1, b2}, (v2f){a1, a2}, s1, s2, prev, c);
                OHS_EQ_FAST_CHUNK(0) OHS_EQ_FAST_CHUNK(1) OHS_EQ_FAST_CHUNK(2) OHS_EQ_FAST_CHUNK(3)
                OHS_EQ_FAST_CHUNK(4) OHS_EQ_FAST_CHUNK(5) OHS_EQ_FAST_CHUNK(6) OHS_EQ_FAST_CHUNK(7)
#undef OHS_EQ_FAST_CHUNK
                eq_wait_loads<8>(nxt);
            }
        } else {
            if (g == g_fast_end && g_fast_end > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const long long xi = ((g + 1) * G + e) * 16 + j;
                nxt[e] = (live && xi < n) ? src[xi] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const long long q = g * G + e;
                if (q < n_chunks) {
                    const long long t0 = q * 16;
                    // conveyor hand-over: drain 16 finished samples y[t0 - 16 - NB + i], refill with inputs
                    const long long yi = t0 - 16 - NB + j;
                    if (live && yi >= 0 && yi < n) dst[yi] = c;
                    c = cur[e];
                    if (t0 >= 16 && t0 + 16 <= n)
                        eq_chunk<SHR, SEL, false>(t0, n, j, en, b0, b1, b2, a1, a2, s1, s2, prev, c);
                    else
                        eq_chunk<SHR, SEL, true>(t0, n, j, en, b0, b1, b2, a1, a2, s1, s2, prev, c);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < G; ++e) cur[e] = nxt[e];
    }
    {
        const long long yi = n_chunks * 16 - 16 - NB + j;
        if (live && yi >= 0 && yi < n) dst[yi] = c;
    }
    if (live && j < NB) {
        state[(cc * 16 + j) * 2 + 0] = s1;
        state[(cc * 16 + j) * 2 + 1] = s2;
    }
}

template <int SHR>
static hipError_t launch_eq_shr(bool sel, int blocks, hipStream_t st, const float *in, float *out,
                                long long ss, long long cs, long long n, int n_chains,
                                const EqPassTable &tab, float *state)
{
    if (sel) hipLaunchKernelGGL((k_eq_pass<SHR, true>), dim3(blocks), dim3(64), 0, st, in, out, ss, cs, n, n_chains, tab, state);
    else     hipLaunchKernelGGL((k_eq_pass<SHR, false>), dim3(blocks), dim3(64), 0, st, in, out, ss, cs, n, n_chains, tab, state);
    return hipGetLastError();
}

hipError_t launch_eq_pass(const float *in, float *out, long long stream_stride, long long ch_stride,
                          long long n, int n_chains, const EqPassTable &tab, int n_bands, float *state,
                          hipStream_t st)
{
    if (n <= 0 || n_chains <= 0) return hipSuccess;
    if (n_bands < 1 || n_bands > 16) return hipErrorInvalidValue;
    const int blocks = (n_chains + 3) / 4;
    bool sel = false;       // some band of this pass disabled -> pass-through select variant
    for (int j = 0; j < n_bands; ++j) sel = sel || !tab.en[j];
#define OHS_EQ_CASE(NBV) \
    case NBV: return launch_eq_shr<16 - NBV>(sel, blocks, st, in, out, stream_stride, ch_stride, n, n_chains, tab, state);
    switch (n_bands) {
        OHS_EQ_CASE(1) OHS_EQ_CASE(2) OHS_EQ_CASE(3) OHS_EQ_CASE(4) OHS_EQ_CASE(5) OHS_EQ_CASE(6)
        OHS_EQ_CASE(7) OHS_EQ_CASE(8) OHS_EQ_CASE(9) OHS_EQ_CASE(10) OHS_EQ_CASE(11) OHS_EQ_CASE(12)
        OHS_EQ_CASE(13) OHS_EQ_CASE(14) OHS_EQ_CASE(15) OHS_EQ_CASE(16)
    }
#undef OHS_EQ_CASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------
// calculate_frequency_response (parametric_eq.rs:190-209), num-complex 0.4.6 op order
// ---------------------------------------------------------------------------------
__global__ void k_eq_freq_response(const float *__restrict__ coeffs, const int *__restrict__ en, int nb,
                                   float fs, const float *__restrict__ freqs, int n, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float pi = 3.14159265358979323846264338327950288f;
    float rr = 1.0f, ri = 0.0f;
    for (int b = 0; b < nb; ++b) {
        if (!en[b]) continue;
        const float b0 = coeffs[5 * b], b1 = coeffs[5 * b + 1], b2 = coeffs[5 * b + 2];
        const float a1 = coeffs[5 * b + 3], a2 = coeffs[5 * b + 4];
        const float omega = 2.0f * pi * freqs[i] / fs;
        const float zr = 1.0f * cosf(-omega), zi = 1.0f * sinf(-omega);      // from_polar(1, -omega)
        const float ns = zr * zr + zi * zi;
        const float ir = zr / ns, ii = -zi / ns;                               // z.powi(-1)
        const float i2r = ir * ir - ii * ii, i2i = ir * ii + ii * ir;          // z.powi(-2)
        const float nr = (b0 + b1 * ir) + b2 * i2r, ni = b1 * ii + b2 * i2i;
        const float dr = (1.0f + a1 * ir) + a2 * i2r, di = a1 * ii + a2 * i2i;
        const float dn = dr * dr + di * di;
        const float qr = (nr * dr + ni * di) / dn, qi = (ni * dr - nr * di) / dn;
        const float tr = rr * qr - ri * qi, ti = rr * qi + ri * qr;
        rr = tr; ri = ti;
    }
    out[i] = hypotf(rr, ri);
}

hipError_t launch_eq_freq_response(const float *d_coeffs, const int *d_en, int nb, float fs,
                                   const float *d_freqs, int n, float *d_out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_eq_freq_response, dim3((n + 63) / 64), dim3(64), 0, st, d_coeffs, d_en, nb, fs,
                       d_freqs, n, d_out);
    return hipGetLastError();
}

}  // namespace ohs
