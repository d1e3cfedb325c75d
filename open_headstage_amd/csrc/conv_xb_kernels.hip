// conv_xb_kernels.hip -- long impulse responses in batch mode, long out-of-place calls: block 8192 / FFT 16384, overlap-save,
// forward transform, product and inverse transform in ONE kernel.
//
// What it replaces: the reference's uniform blocking (src/dsp/convolution.rs:22-23: block 512 / FFT 1024; :120-132: one
// spectrum per 512 taps; :261-273: P x 1024 complex products per block).  The block-2048 plan (conv_lb_kernels.hip) halved
// BASELINE config 4 but still writes every window's spectrum to a ring and reads it back (445 MB per step through HBM for
// 134 MB of audio).  With 8192-tap partitions a 16 384-tap response has TWO, and a spectrum need never leave the workgroup
// that computed it:
//     W_i = Z_i CD_0 + carry ;  carry' = Z_i CD_1          (Z_i = the spectrum of window i = blocks [i - 1, i])
// -- the second partition's product is computed one block early and waits in registers.  A workgroup runs consecutive
// blocks of one stream (the run's first window is transformed for its carry alone); per frame 8 B in (+ the window's
// first half again: the output is stored non-temporally so that it stays in the L2), 8 B out, nothing else.  The state at
// rest is the block-2048 plan's: the streams' last input frames (ConvState::d_xhist), which the kernel appends to itself.
//
// The 16 384-point transform = sixteen of the wave's 1024-point transforms (wave_fft.hpp) plus a lane-local radix-16 step
// (tools/proto_xb_conv.py is the index-exact CPU model):
//   forward, decimation in frequency:  v_c[n1] = w16384^(n1 c) sum_n2 w16^(n2 c) z[n1 + 1024 n2];  Z[16 k1 + c] = FFT1024(v_c)[k1]
//   inverse, decimation in time:       y_c = IFFT1024(W[16 k1 + c]);  x[n1 + 1024 n2] = sum_c w16^(-n2 c) conj(w16384^(n1 c)) y_c[n1]
// of which only n2 = 8 .. 15 -- the window's second half -- is stored.
//
// EIGHT waves; wave w transforms the classes c = w AND 16 - w (wave 0: 0 and 8).  Bin k = 16 k1 + c has its mirror
// N - k = 16 (1023 - k1) + (16 - c) in the OTHER class of the same wave, at lane 63 - L and register 15 - R of the 1024-point
// transform's bin layout (class 0 mirrors into itself by (1024 - k1) & 1023: wave_mirror; class 8 into itself by the
// reversal): the packed two-channel product W[k] = Z[k] C[k] + conj(Z[N - k]) D[k] (C = (A - jB)/2, D = (A + jB)/2,
// A = H_lsl + j H_lsr, B = H_rsl + j H_rsr) finds its second input by a lane reversal (ds_bpermute) inside the wave -- no pair
// image, no workgroup barrier between the forward transforms and the inverse ones -- and the tables are stored per bin in
// the transforms' own layout ([partition][wave][class][register][lane], built by this kernel's tables mode).
//
// LDS of a workgroup: the 8.5 KiB twiddle tables + sixteen slabs of 1088 complex values (144.5 KiB: one workgroup per CU).
// The slabs are the waves' transform scratch AND the exchange image X1 between the radix-16 steps (thread = the samples
// n1 = 2 tid and 2 tid + 1 of every sixteenth: 8-byte global, 16-byte LDS accesses) and the transforms (wave = class):
//   X1   class c, sample n1  at  1088 c + n1          -- contiguous for both sides
// Three workgroup barriers per block.  256 VGPRs (two partitions: 243 used, no scratch).
//
// Measured (profiles/r05_xb_*, LABNOTES.md round 5): BASELINE config 4 0.118 ms per step against 0.135-0.140 for block 2048
// (0.152-0.165 for a first form with sixteen waves and a pair image in LDS: experiments/conv_xb16_kernels.hip); 1024 blocks
// per call 0.38 against 0.55; 256 streams 0.41 against 0.56.  What bounds it: one workgroup per CU means the CUs' window
// loads come as bursts (18-27 us of a step are pure load time); the vector unit is ~40 % busy.
#include "kernels.h"
#include "wave_fft.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>

namespace ohs {

typedef float xb_v2f __attribute__((ext_vector_type(2)));
typedef float xb_v4f __attribute__((ext_vector_type(4)));
typedef unsigned xb_v2u __attribute__((vector_size(8)));

constexpr int kXbFft = 16384;
constexpr int kXbSlab = 1088;
constexpr size_t kXbLdsBytes = (kTabComplex + 16 * (size_t)kXbSlab) * sizeof(float2);
static_assert(kXbSlab >= kWaveLdsComplex && kXbSlab >= 1024, "LDS plan");

// host: w16384^(n1 c), c = 1 .. 15, from f64
void conv_xb_build_twiddles(float2 *out)
{
    for (int c = 1; c < 16; ++c)
        for (int n1 = 0; n1 < 1024; ++n1) {
            const double ang = -2.0 * 3.14159265358979323846264338327950288 * (double)(n1 * c) / 16384.0;
            out[(c - 1) * 1024 + n1] = make_float2((float)cos(ang), (float)sin(ang));
        }
}

// workgroup -> (stream, run) with all runs of a stream on ONE XCD (conv_lb_kernels.hip: lb_job)
__device__ __forceinline__ bool xb_job(int per_stream, int n_streams, size_t &s, int &idx)
{
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    s = (size_t)(slot / (unsigned)per_stream) * 8 + xcd;
    idx = (int)(slot % (unsigned)per_stream);
    return s < (size_t)n_streams;
}
static unsigned xb_grid(int per_stream, int n_streams) { return (unsigned)(((n_streams + 7) / 8) * 8 * per_stream); }

// acc += z c + conj(m) d on (re, im) register pairs (conv_lb_kernels.hip: lb_cmac)
__device__ __forceinline__ void xb_cmac(xb_v2f &acc, xb_v2f z, xb_v2f c, xb_v2f m, xb_v2f d)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(z), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(acc) : "v"(z), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(m), "v"(d));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "+v"(acc) : "v"(m), "v"(d));
}

__device__ __forceinline__ float2 xc_rev(float2 v, int src)
{
    float2 o;
    o.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.x)));
    o.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.y)));
    return o;
}

#ifndef XB_ST_AUX
#define XB_ST_AUX 2       // nt: the output streams past the L2, the window's first half (read again one block later) stays
#endif
template <int P2>
__global__ __launch_bounds__(512) void k_conv_xb(const ConvXbArgs A)
{
    ohs_set_fp_mode(A.fp_mode);
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    float2 *X = smem + kTabComplex;
    const int tid = threadIdx.x;
    fill_twiddle_tables(tab, A.tw, tid, 512);
    const int runs = (A.n_blk + A.run - 1) / A.run;
    size_t s;
    int chunk;
    if (!xb_job(runs, A.n_streams, s, chunk)) return;
    const int i0 = chunk * A.run, i1 = i0 + A.run < A.n_blk ? i0 + A.run : A.n_blk;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int cA = w, cB = w == 0 ? 8 : 16 - w;
    float2 *slabA = X + kXbSlab * cA, *slabB = X + kXbSlab * cB;
    const int rev = (63 - lane) << 2;
    // One workgroup per CU, all started together: without this every CU would load its window in the same moment and compute in
    // the same moment.  Every second workgroup starts A.stagger x 3.5 us late (config 4: 0.122 -> 0.118 ms at 1; 2: the same; 3: worse).
    if ((blockIdx.x >> 3) & 1)
        for (int k = 0; k < A.stagger; ++k) __builtin_amdgcn_s_sleep(127);

    const long long L2 = 2 * A.xh_len;
    const int io_bytes = (int)(A.io_frames * 4);
    const __amdgpu_buffer_rsrc_t r_in_l = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(A.in) + s * A.in_stream_stride, 0, io_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_in_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(A.in) + s * A.in_stream_stride + A.in_ch_stride, 0, io_bytes, 0x00020000);
    const float *xh_base = A.xhist ? A.xhist + s * 2 * L2 : A.in;
    const __amdgpu_buffer_rsrc_t r_xh_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xh_base), 0, (int)(L2 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_xh_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xh_base) + (A.xhist ? L2 : 0), 0,
                                                                            (int)(L2 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_tw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(A.tw16384), 0, 15 * 1024 * 8, 0x00020000);
    // The radix-16 steps: thread = the samples n1 = 2 tid and 2 tid + 1 of every sixteenth -- 8-byte global accesses, 16-byte
    // twiddle loads and 16-byte X1 accesses serve both.
    const int v8 = tid * 8, v16 = tid * 16, l16 = lane * 16, v8h = (tid & 255) * 8;
    const int half = 512 * (w >> 2);                    // the wave's 128 frames sit in this half of a 1024-frame sixteenth
#ifdef OHS_EXPERIMENTS
    const int dskip = A.debug_skip;     // timing knock-outs and the start stagger override (Tuning::xb_skip)
#else
    constexpr int dskip = 0;
#endif
    auto twiddle2 = [&](int cc, float2 &t0, float2 &t1) {      // w16384^((2 tid + e) cc), cc = 1 .. 15
        if (dskip & 2) { t0 = make_float2(1.0f, 0.001f * cc); t1 = t0; return; }
        const xb_v4f t = __builtin_bit_cast(xb_v4f, __builtin_amdgcn_raw_buffer_load_b128(r_tw, v16, (cc - 1) * 8192, 0));
        t0 = make_float2(t.x, t.y); t1 = make_float2(t.z, t.w);
    };
    // one sixteenth (1024 frames from frame f0 of the call's axis) of a window: the thread's frames f0 + 2 tid, + 1
    // own: the sixteenth belongs to the block's own frames (the window's second half) -- every frame of the call is loaded as
    // such exactly once, and the call's last xh_cnt frames join the input history then, in the half of the ring no window reads
    const long long app0 = A.io_frames - A.xh_cnt;
    auto load16th = [&](long long f0, bool own, xb_v2f &l, xb_v2f &r) {
        l = (xb_v2f){0.f, 0.f}; r = (xb_v2f){0.f, 0.f};
        if (dskip & 1) return;
        if (f0 >= 0) {
            if (f0 + half < A.io_frames) {              // (uniform over the wave: everything is a multiple of 512)
                const int so = __builtin_amdgcn_readfirstlane((int)(f0 * 4));
                l = __builtin_bit_cast(xb_v2f, __builtin_amdgcn_raw_buffer_load_b64(r_in_l, v8, so, 0));
                r = __builtin_bit_cast(xb_v2f, __builtin_amdgcn_raw_buffer_load_b64(r_in_r, v8, so, 0));
                if (own && A.xh_append && f0 + half >= app0) {
                    long long pos = A.xh_head + (f0 + half - app0);     // (a 512-frame chunk does not straddle the ring's wrap)
                    pos = pos >= L2 ? pos - L2 : pos;
                    const int sa = __builtin_amdgcn_readfirstlane((int)(pos * 4));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xb_v2u, l), r_xh_l, v8h, sa, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xb_v2u, r), r_xh_r, v8h, sa, 0);
                }
            }
        } else if (A.xhist && A.xh_len + f0 >= 0) {
            long long idx = A.xh_head + f0 + half;      // (the head is a multiple of 512: a half sixteenth does not straddle the wrap)
            idx = idx < 0 ? idx + L2 : (idx >= L2 ? idx - L2 : idx);
            const int so = __builtin_amdgcn_readfirstlane((int)(idx * 4));
            l = __builtin_bit_cast(xb_v2f, __builtin_amdgcn_raw_buffer_load_b64(r_xh_l, v8h, so, 0));
            r = __builtin_bit_cast(xb_v2f, __builtin_amdgcn_raw_buffer_load_b64(r_xh_r, v8h, so, 0));
        }
    };
    // window i -> the spectra of the wave's two classes in the 1024-point transform's bin layout
    auto forward = [&](int i, bool main_block, float2 (&ZA)[16], float2 (&ZB)[16]) {
        const long long w0 = A.tables_mode ? (long long)i * 8192 : ((long long)i - 1) * 8192;
        float2 z0[16], z1[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) {
            xb_v2f l = {0.f, 0.f}, r = {0.f, 0.f};
            if (!A.tables_mode || n2 < 8) load16th(w0 + 1024 * n2, main_block && n2 >= 8 && !A.tables_mode, l, r);
            z0[n2] = make_float2(l.x, r.x); z1[n2] = make_float2(l.y, r.y);
        }
        float2 t0[16], t1[16];
#pragma unroll
        for (int cc = 1; cc < 16; ++cc) twiddle2(cc, t0[cc], t1[cc]);       // (requested with the frames: one wait for both)
        dft16<false>(z0);
#pragma unroll
        for (int cc = 1; cc < 16; ++cc) z0[cc] = cmul(z0[cc], t0[cc]);
        dft16<false>(z1);
#pragma unroll
        for (int cc = 1; cc < 16; ++cc) z1[cc] = cmul(z1[cc], t1[cc]);
        __syncthreads();                                // whoever still reads X1 is done
#pragma unroll
        for (int cc = 0; cc < 16; ++cc)
            *(volatile __attribute__((address_space(3))) lds_u32x4 *)&X[kXbSlab * cc + 2 * tid] =
                (lds_u32x4){__float_as_uint(z0[cc].x), __float_as_uint(z0[cc].y), __float_as_uint(z1[cc].x), __float_as_uint(z1[cc].y)};
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 16; ++a) { ZA[a] = lds_ld(&slabA[64 * a + lane]); ZB[a] = lds_ld(&slabB[64 * a + lane]); }
        wave_lds_sync();
        if (!(dskip & 4)) {
            wave_fft_fwd(ZA, slabA, tab, lane);         // register 4 r + j holds k1 = 4 lane + j + 256 r, i.e. bin 16 k1 + cA
            wave_fft_fwd(ZB, slabB, tab, lane);
        }
    };
    const __amdgpu_buffer_rsrc_t r_cd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(A.cd), 0, P2 * kXbFft * 16, 0x00020000);
    auto table = [&](int part, int h, int R) {      // (C[k], D[k]) of the bin in register R of class h (0: cA, 1: cB)
        if (dskip & 8) return (xb_v4f){0.5f, 0.25f, 0.125f, 0.0625f};
        return __builtin_bit_cast(xb_v4f, __builtin_amdgcn_raw_buffer_load_b128(r_cd, l16, ((((part * 8 + w) * 2 + h) * 16 + R) * 64) * 16, 0));
    };
    auto v2 = [](float2 a) { return (xb_v2f){a.x, a.y}; };
    // acc = carry + Z CD_0 ; carry = Z CD_1, register pairs (R, 15 - R) of both classes together: their mirrors are each other's
    // reversals.  WA / WB receive acc; ZA / ZB are consumed, CA / CB are read (old carry) and rewritten (new carry).
    auto products = [&](float2 (&ZA)[16], float2 (&ZB)[16], xb_v2f (&WA)[16], xb_v2f (&WB)[16], xb_v2f (&CA)[16], xb_v2f (&CB)[16],
                        bool first, bool last) {
        if (w == 0) {       // class 0 mirrors into itself by k1 <-> (1024 - k1) & 1023: fetched once, parked in the wave's own slab
            float2 zmA[16];
            wave_mirror(ZA, zmA, lane);
#pragma unroll
            for (int R = 0; R < 16; ++R) lds_st(&slabA[64 * R + lane], zmA[R]);
        }
        xb_v4f tn[2][2][P2];                            // [R or Q][class][partition]: the NEXT register pair's rows
        auto request = [&](int R, xb_v4f (&t)[2][2][P2]) {
#pragma unroll
            for (int part = 0; part < P2; ++part) {
                t[0][0][part] = table(part, 0, R); t[1][0][part] = table(part, 0, 15 - R);
                t[0][1][part] = table(part, 1, R); t[1][1][part] = table(part, 1, 15 - R);
            }
        };
        request(0, tn);
#pragma unroll
        for (int R = 0; R < 8; ++R) {
            const int Q = 15 - R;
            xb_v4f t[2][2][P2];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int part = 0; part < P2; ++part) t[x][h][part] = tn[x][h][part];
            __builtin_amdgcn_sched_barrier(0);
            if (R + 1 < 8) request(R + 1, tn);
            const float2 a0 = ZA[R], a1 = ZA[Q], b0 = ZB[R], b1 = ZB[Q];
            float2 mA0, mA1, mB0, mB1;
            if (w == 0) {
                mA0 = lds_ld(&slabA[64 * R + lane]); mA1 = lds_ld(&slabA[64 * Q + lane]);
                mB0 = xc_rev(b1, rev); mB1 = xc_rev(b0, rev);      // class 8 mirrors into itself by the reversal
            } else {
                mA0 = xc_rev(b1, rev); mA1 = xc_rev(b0, rev);
                mB0 = xc_rev(a1, rev); mB1 = xc_rev(a0, rev);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!first) {
                WA[R] = CA[R]; WA[Q] = CA[Q]; WB[R] = CB[R]; WB[Q] = CB[Q];
                xb_cmac(WA[R], v2(a0), t[0][0][0].lo, v2(mA0), t[0][0][0].hi);
                xb_cmac(WA[Q], v2(a1), t[1][0][0].lo, v2(mA1), t[1][0][0].hi);
                xb_cmac(WB[R], v2(b0), t[0][1][0].lo, v2(mB0), t[0][1][0].hi);
                xb_cmac(WB[Q], v2(b1), t[1][1][0].lo, v2(mB1), t[1][1][0].hi);
            }
            if constexpr (P2 == 2) {
                CA[R] = (xb_v2f){0.f, 0.f}; CA[Q] = (xb_v2f){0.f, 0.f}; CB[R] = (xb_v2f){0.f, 0.f}; CB[Q] = (xb_v2f){0.f, 0.f};
                if (!last) {
                    xb_cmac(CA[R], v2(a0), t[0][0][1].lo, v2(mA0), t[0][0][1].hi);
                    xb_cmac(CA[Q], v2(a1), t[1][0][1].lo, v2(mA1), t[1][0][1].hi);
                    xb_cmac(CB[R], v2(b0), t[0][1][1].lo, v2(mB0), t[0][1][1].hi);
                    xb_cmac(CB[Q], v2(b1), t[1][1][1].lo, v2(mB1), t[1][1][1].hi);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (A.tables_mode) {
        for (int i = i0; i < i1; ++i) {
            float2 ZA[16], ZB[16];
            forward(i, false, ZA, ZB);
            float2 *dst = reinterpret_cast<float2 *>(A.ab) + ((size_t)s * A.n_blk + i) * kXbFft + (size_t)(w * 2) * 16 * 64 + lane;
#pragma unroll
            for (int R = 0; R < 16; ++R) { dst[R * 64] = ZA[R]; dst[(16 + R) * 64] = ZB[R]; }
        }
        return;
    }

    xb_v2f CA[16], CB[16], WA[16], WB[16];
#pragma unroll
    for (int R = 0; R < 16; ++R) { CA[R] = (xb_v2f){0.f, 0.f}; CB[R] = (xb_v2f){0.f, 0.f}; }
    if constexpr (P2 == 2) {
        float2 ZA[16], ZB[16];
        forward(i0 - 1, false, ZA, ZB);
        products(ZA, ZB, WA, WB, CA, CB, true, false);
    }
    const __amdgpu_buffer_rsrc_t r_out_l = __builtin_amdgcn_make_buffer_rsrc(A.out + s * A.out_stream_stride, 0, io_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out_r = __builtin_amdgcn_make_buffer_rsrc(A.out + s * A.out_stream_stride + A.out_ch_stride, 0,
                                                                             io_bytes, 0x00020000);
    for (int i = i0; i < i1; ++i) {
        {
            float2 ZA[16], ZB[16];
            forward(i, true, ZA, ZB);
            products(ZA, ZB, WA, WB, CA, CB, false, i + 1 >= i1);
        }
        {
            float2 v[16];
#pragma unroll
            for (int R = 0; R < 16; ++R) v[R] = make_float2(WA[R].x, WA[R].y);
            if (!(dskip & 16)) wave_fft_inv(v, slabA, tab, lane);          // register a holds y_cA[64 a + lane]
#pragma unroll
            for (int a = 0; a < 16; ++a) lds_st(&slabA[64 * a + lane], v[a]);
#pragma unroll
            for (int R = 0; R < 16; ++R) v[R] = make_float2(WB[R].x, WB[R].y);
            if (!(dskip & 16)) wave_fft_inv(v, slabB, tab, lane);
#pragma unroll
            for (int a = 0; a < 16; ++a) lds_st(&slabB[64 * a + lane], v[a]);
        }
        float2 t0[16], t1[16];
#pragma unroll
        for (int cc = 1; cc < 16; ++cc) twiddle2(cc, t0[cc], t1[cc]);       // (in flight across the barrier)
        __syncthreads();
        float2 y0[16], y1[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) lds_ld2(reinterpret_cast<const float4 *>(&X[kXbSlab * cc + 2 * tid]), y0[cc], y1[cc]);
#pragma unroll
        for (int cc = 1; cc < 16; ++cc) { y0[cc] = cmulc(y0[cc], t0[cc]); y1[cc] = cmulc(y1[cc], t1[cc]); }
        dft16<true>(y0);
        dft16<true>(y1);
#pragma unroll
        for (int n2 = 8; n2 < 16; ++n2) {
            const long long f0 = (long long)i * 8192 + 1024 * (n2 - 8);
            if (f0 + half < A.io_frames && !(dskip & 32)) {     // (uniform over the wave)
                const int so = __builtin_amdgcn_readfirstlane((int)(f0 * 4));
                const xb_v2f ol = {y0[n2].x * A.gain, y1[n2].x * A.gain}, orr = {y0[n2].y * A.gain, y1[n2].y * A.gain};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xb_v2u, ol), r_out_l, v8, so, XB_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xb_v2u, orr), r_out_r, v8, so, XB_ST_AUX);
            }
        }
        // (the next forward() starts with a barrier: X1 has been read by then)
    }
}

// (C, D) tables per bin, in the transforms' own layout, from the spectra of A = h_lsl + j h_lsr (ab[0][i]) and B = h_rsl + j h_rsr (ab[1][i])
__global__ __launch_bounds__(256) void k_conv_xb_build_cd(const float2 *__restrict__ ab, int P2, float4 *__restrict__ cd)
{
    const int i = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;      // q < 16384
    const float2 a = ab[((size_t)0 * P2 + i) * kXbFft + q];
    const float2 b = ab[((size_t)1 * P2 + i) * kXbFft + q];
    const float h = 0.5f / (float)kXbFft;
    cd[(size_t)i * kXbFft + q] = make_float4(h * (a.x + b.y), h * (a.y - b.x), h * (a.x - b.y), h * (a.y + b.x));
}

static hipError_t xb_allow_lds(const void *fn, std::atomic<unsigned long long> &done_mask)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done_mask.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kXbLdsBytes);
    if (e == hipSuccess) done_mask.fetch_or(bit, std::memory_order_release);
    return e;
}

// blocks per workgroup: one workgroup per CU at a time; a run costs its blocks + half a block for the carry's window
int conv_xb_run_for(int n_streams, int n_blk, int P2, int num_cus)
{
    int best_run = 1;
    double best = -1.0;
    for (int run = 1; run <= n_blk && run <= 256; ++run) {      // (the search stops at 256 blocks per workgroup)
        const long long wgs = (long long)n_streams * ((n_blk + run - 1) / run);
        const long long rounds = (wgs + num_cus - 1) / num_cus;
        const double cost = (double)rounds * (run + (P2 == 2 ? 0.5 : 0.0));
        if (best < 0.0 || cost < best - 1e-9) { best = cost; best_run = run; }
    }
    return best_run;
}

hipError_t launch_conv_xb(const ConvXbArgs &a, hipStream_t st)
{
    if (a.n_streams <= 0 || a.n_blk <= 0 || a.run <= 0 || !a.in || !a.tw || !a.tw16384 || (a.P2 != 1 && a.P2 != 2))
        return hipErrorInvalidValue;
    if (a.tables_mode ? !a.ab : (!a.out || !a.cd)) return hipErrorInvalidValue;
    if (a.io_frames < 0 || (a.io_frames & 511) || a.stagger < 0 || a.stagger > 64 ||
        (a.xh_append && (a.xh_append != a.xhist || a.xh_cnt < 0 || (a.xh_cnt & 511) || a.xh_cnt > a.xh_len || a.xh_cnt > a.io_frames)) ||
        (a.xhist && (a.xh_len <= 0 || (a.xh_len & 511) || a.xh_head < 0 || a.xh_head >= 2 * a.xh_len)))
        return hipErrorInvalidValue;
    static std::atomic<unsigned long long> ok1{0}, ok2{0};
    const int runs = (a.n_blk + a.run - 1) / a.run;
    const unsigned grid = xb_grid(runs, a.n_streams);
    hipError_t e;
    if (a.P2 == 2 && !a.tables_mode) {
        e = xb_allow_lds(reinterpret_cast<const void *>(&k_conv_xb<2>), ok2);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_conv_xb<2>, dim3(grid), dim3(512), kXbLdsBytes, st, a);
    } else {
        e = xb_allow_lds(reinterpret_cast<const void *>(&k_conv_xb<1>), ok1);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_conv_xb<1>, dim3(grid), dim3(512), kXbLdsBytes, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_conv_xb_build_cd(const float2 *ab, int P2, float4 *cd, hipStream_t st)
{
    if (!ab || !cd || P2 < 1 || P2 > 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv_xb_build_cd, dim3(kXbFft / 256, P2), dim3(256), 0, st, ab, P2, cd);
    return hipGetLastError();
}

}  // namespace ohs
