// conv_xb_kernels.hip -- long impulse responses in batch mode, long calls: block 8192 / FFT 16384, overlap-save, ONE kernel.
//
// EXPERIMENTS BUILD ONLY (libohs_hip_exp.so; off unless ohs_debug_set_tuning("xb_min_p", ..) turns it on): parity-green
// (tests/test_gpu_conv_xb.py), 356 MB of measured traffic per step against the block-2048 plan's 445, but no faster than it on BASELINE
// config 4: the sixteen-wave form below 0.152-0.165 ms against 0.133-0.145, the eight-wave form at the end of this file 0.130 (0.104
// against 0.120 at 8192 taps) -- for the reasons at the end of this comment.  Kept as the measured "no" it is (LABNOTES.md).
//
// What it replaces: the reference's uniform blocking (src/dsp/convolution.rs:22-23: block 512 / FFT 1024; :120-132: one
// spectrum per 512 taps; :261-273: P x 1024 complex products per block).  The block-2048 plan (conv_lb_kernels.hip) halved
// BASELINE config 4 but still writes every window's spectrum to a ring and reads it back (48 B per frame through HBM
// against 16 B of audio).  With 8192-tap partitions a 16 384-tap response has TWO, and a spectrum need never leave the
// workgroup that computed it:
//     W_i = Z_i CD_0 + carry ;  carry' = Z_i CD_1          (Z_i = the spectrum of window i = blocks [i - 1, i])
// -- the second partition's product is computed one block early and waits in registers.  A workgroup runs consecutive
// blocks of one stream (the run's first window is transformed for its carry alone); per frame 8 B in (+ the window's
// first half again, an L2 hit: the same workgroup loaded it one block earlier), 8 B out, nothing else.  The state at rest
// is the block-2048 plan's: the streams' last input frames (ConvState::d_xhist).
//
// The 16 384-point transform = sixteen of the wave's 1024-point transforms (wave_fft.hpp), one per wave of a 16-wave
// workgroup, plus a lane-local radix-16 step (tools/proto_xb_conv.py is the index-exact CPU model):
//   forward, decimation in frequency:  v_c[n1] = w16384^(n1 c) sum_n2 w16^(n2 c) z[n1 + 1024 n2];  Z[16 k1 + c] = FFT1024(v_c)[k1]
//   inverse, decimation in time:       y_c = IFFT1024(W[16 k1 + c]);  x[n1 + 1024 n2] = sum_c w16^(-n2 c) conj(w16384^(n1 c)) y_c[n1]
// of which only n2 = 8 .. 15 -- the window's second half -- is stored.
//
// Spectra are handled as PAIRS: slot q = (Z[q], Z[16384 - q]), slot 0 = (Z[0], Z[8192]) (kernels.h: the packed two-channel
// product finds both of its inputs in one slot).
//
// LDS of a workgroup: the 8.5 KiB twiddle tables + sixteen slabs of 1088 complex values (144.5 KiB in all; one workgroup
// per CU).  The slabs are the waves' transform scratch AND the two exchange images (never at the same time):
//   X1   class c, sample n1  at  1088 c + n1                                   thread (n1 = tid) <-> wave c (n1 = 64 a + lane)
//   XP   bin k               at  8704 (k >= 8192) + 65 (q >> 6) + (q & 63),  q = k < 8192 ? k : (16384 - k) & 8191
//                                                                               wave c (k = 64 lane + ..) <-> thread (q = tid + 1024 p)
// Both sides of both images are conflict-free (lane stride 1, or 65 complex = 130 dwords = 2 mod 32).
//
// Why it loses (profiles/r05_xb_*): 144.5 KiB of LDS means ONE workgroup per CU, and its sixteen waves walk through the block's
// phases -- load + radix-16, X1, 1024-point transforms, XP, products (512 KiB of table rows per block through the CU's 64 B / clk
// vector-memory path), XP, transforms, X1, radix-16 + store -- between eight workgroup barriers, so vector unit (39 % busy),
// LDS and memory take turns instead of overlapping as they do where two or three workgroups share a CU: 33 us per block where
// the units' own times would allow 15-20.  The 128-register budget of sixteen waves leaves nothing to prefetch into.
#include "../kernels.h"
#include "../wave_fft.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>

namespace ohs {

typedef float xb_v2f __attribute__((ext_vector_type(2)));
typedef float xb_v4f __attribute__((ext_vector_type(4)));
typedef unsigned xb_v2u __attribute__((vector_size(8)));

constexpr int kXbFft = 16384;
constexpr int kXbPairs = 8192;
constexpr int kXbSlab = 1088;
constexpr int kXbPlane = 8 * kXbSlab;           // 8704 >= 128 rows x 65
constexpr size_t kXbLdsBytes = (kTabComplex + 16 * (size_t)kXbSlab) * sizeof(float2);
static_assert(kXbSlab >= kWaveLdsComplex && kXbSlab >= 1024 && 65 * 128 <= kXbPlane, "LDS plan");

__device__ __forceinline__ int xb_pair_addr(int k)
{
    const int hi = k >= kXbPairs;
    const int q = hi ? ((kXbFft - k) & (kXbPairs - 1)) : k;
    return kXbPlane * hi + 65 * (q >> 6) + (q & 63);
}

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

// P2: partitions of 8192 taps (1 or 2).  One workgroup = 16 waves = (stream, run of A.run consecutive blocks).
template <int P2>
__global__ __launch_bounds__(1024) void k_conv_xb(const ConvXbArgs A)
{
    ohs_set_fp_mode(A.fp_mode);
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    float2 *X = smem + kTabComplex;
    const int tid = threadIdx.x;
    fill_twiddle_tables(tab, A.tw, tid, 1024);
    const int runs = (A.n_blk + A.run - 1) / A.run;
    size_t s;
    int chunk;
    if (!xb_job(runs, A.n_streams, s, chunk)) return;
    const int i0 = chunk * A.run, i1 = i0 + A.run < A.n_blk ? i0 + A.run : A.n_blk;
    const int c = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    float2 *slab = X + kXbSlab * c;
    const int pbase = 65 * (tid >> 6) + lane;            // xb_pair_addr(tid + 1024 p) = pbase + 1040 p (+ plane)

    // Every global access is a buffer access: the thread's bytes in the vector offset, block / sixteenth / partition / pair group in
    // the SCALAR offset (no 64-bit address registers -- the 128-register budget of a 16-wave workgroup has none to spare).  The
    // scalar offset takes no part in the resource's bounds check: frames behind the call's end are fenced off by wave-uniform branches.
    const long long L2 = 2 * A.xh_len;
    const int io_bytes = (int)(A.io_frames * 4);
    const __amdgpu_buffer_rsrc_t r_in_l = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(A.in) + s * A.in_stream_stride, 0, io_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_in_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(A.in) + s * A.in_stream_stride + A.in_ch_stride, 0, io_bytes, 0x00020000);
    const float *xh_base = A.xhist ? A.xhist + s * 2 * L2 : A.in;         // (never read when xhist is NULL)
    const __amdgpu_buffer_rsrc_t r_xh_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xh_base), 0, (int)(L2 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_xh_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xh_base) + (A.xhist ? L2 : 0), 0,
                                                                            (int)(L2 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_tw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(A.tw16384), 0, 15 * 1024 * 8, 0x00020000);
    const int v4 = tid * 4, v8 = tid * 8, v16 = tid * 16, v4h = (tid & 511) * 4;
    const int half = 512 * (c >> 3);                    // the wave's 64 frames sit in this half of a 1024-frame sixteenth
    const int dskip = A.debug_skip;
    auto twiddle = [&](int cc) {            // w16384^(tid cc), cc = 1 .. 15
        if (dskip & 2) return make_float2(1.0f, 0.001f * cc);
        return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r_tw, v8, (cc - 1) * 8192, 0));
    };

    // window i = blocks [i - 1, i] of the call (tables mode: [block i, 8192 zeros]) -> its spectrum as the thread's 8 pairs.
    // Frames in front of the call come from the input history.  (The source of a sixteenth is uniform over the workgroup:
    // sixteenths are 1024 frames, the call starts at frame 0.)
    auto forward = [&](int i, xb_v4f (&zp)[8]) {
        float2 z[16];
        const long long w0 = A.tables_mode ? (long long)i * 8192 : ((long long)i - 1) * 8192;       // the window's first frame
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) {
            float l = 0.f, r = 0.f;
            const long long f0 = w0 + 1024 * n2;
            if ((!A.tables_mode || n2 < 8) && !(dskip & 1)) {
                if (f0 >= 0) {
                    if (f0 + half < A.io_frames) {              // (uniform over the wave: everything is a multiple of 512)
                        const int so = __builtin_amdgcn_readfirstlane((int)(f0 * 4));
                        l = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_in_l, v4, so, 0));
                        r = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_in_r, v4, so, 0));
                    }
                } else if (A.xhist && A.xh_len + f0 >= 0) {
                    long long idx = A.xh_head + f0 + half;      // (the head is a multiple of 512: a half sixteenth does not straddle the wrap)
                    idx = idx < 0 ? idx + L2 : (idx >= L2 ? idx - L2 : idx);
                    const int so = __builtin_amdgcn_readfirstlane((int)(idx * 4));
                    l = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_xh_l, v4h, so, 0));
                    r = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_xh_r, v4h, so, 0));
                }
            }
            z[n2] = make_float2(l, r);
        }
        if (!(dskip & 64)) dft16<false>(z);
#pragma unroll
        for (int cc = 1; cc < 16; ++cc) z[cc] = cmul(z[cc], twiddle(cc));
        __syncthreads();                                // whoever still reads the slabs' previous image is done
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) lds_st(&X[kXbSlab * cc + tid], z[cc]);
        __syncthreads();
        float2 v[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = lds_ld(&slab[64 * a + lane]);
        wave_lds_sync();
        if (!(dskip & 4)) wave_fft_fwd(v, slab, tab, lane);               // register 4 r + j holds k1 = 4 lane + j + 256 r, i.e. bin 16 k1 + c
        __syncthreads();                                // every wave is done with its slab: the slabs become the XP image
#pragma unroll
        for (int R = 0; R < 16; ++R) lds_st(&X[xb_pair_addr(64 * lane + 16 * (R & 3) + 4096 * (R >> 2) + c)], v[R]);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float2 a = lds_ld(&X[pbase + 1040 * p]), b = lds_ld(&X[kXbPlane + pbase + 1040 * p]);
            zp[p] = (xb_v4f){a.x, a.y, b.x, b.y};
        }
    };
    // The thread's pairs against the tables: acc = carry + Z CD_0 ; carry = Z CD_1 (FIRST: carry = Z CD_1 only -- the run's
    // first window; LAST: no carry is needed behind the run's last block).  Two pairs' table rows in flight at a time.
    const bool special = tid == 0;                      // pair slot 0 holds bins 0 and 8192, each its own mirror
    const __amdgpu_buffer_rsrc_t r_cd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(A.cd), 0, P2 * 2 * kXbPairs * 16, 0x00020000);
    auto table = [&](int part, int plane, int p) {
        if (dskip & 8) return (xb_v4f){0.5f, 0.25f, 0.125f, 0.0625f};
        return __builtin_bit_cast(xb_v4f, __builtin_amdgcn_raw_buffer_load_b128(r_cd, v16, ((part * 2 + plane) * kXbPairs + 1024 * p) * 16, 0));
    };
    auto products = [&](const xb_v4f (&zp)[8], xb_v2f (&acc)[8][2], xb_v2f (&carry)[8][2], bool first, bool last) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            xb_v4f t[2][P2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int part = 0; part < P2; ++part)
#pragma unroll
                    for (int plane = 0; plane < 2; ++plane) t[e][part][plane] = table(part, plane, 2 * g + e);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int p = 2 * g + e;
                const xb_v4f zz = zp[p];
                xb_v2f m0 = zz.hi, m1 = zz.lo;
                if (p == 0) { m0 = special ? zz.lo : m0; m1 = special ? zz.hi : m1; }
                if (!first) {
                    acc[p][0] = carry[p][0]; acc[p][1] = carry[p][1];
                    xb_cmac(acc[p][0], zz.lo, t[e][0][0].lo, m0, t[e][0][0].hi);
                    xb_cmac(acc[p][1], zz.hi, t[e][0][1].lo, m1, t[e][0][1].hi);
                }
                if constexpr (P2 == 2) {
                    carry[p][0] = (xb_v2f){0.f, 0.f}; carry[p][1] = (xb_v2f){0.f, 0.f};
                    if (!last) {
                        xb_cmac(carry[p][0], zz.lo, t[e][1][0].lo, m0, t[e][1][0].hi);
                        xb_cmac(carry[p][1], zz.hi, t[e][1][1].lo, m1, t[e][1][1].hi);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (A.tables_mode) {
        for (int i = i0; i < i1; ++i) {
            xb_v4f zp[8];
            forward(i, zp);
            xb_v4f *dst = reinterpret_cast<xb_v4f *>(A.ab) + ((size_t)s * A.n_blk + i) * kXbPairs + tid;
#pragma unroll
            for (int p = 0; p < 8; ++p) dst[1024 * p] = zp[p];
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }

    xb_v2f carry[8][2], acc[8][2];
#pragma unroll
    for (int p = 0; p < 8; ++p) { carry[p][0] = (xb_v2f){0.f, 0.f}; carry[p][1] = (xb_v2f){0.f, 0.f}; }
    if constexpr (P2 == 2) {
        xb_v4f zp[8];
        forward(i0 - 1, zp);
        products(zp, acc, carry, true, false);
    }
    const __amdgpu_buffer_rsrc_t r_out_l = __builtin_amdgcn_make_buffer_rsrc(A.out + s * A.out_stream_stride, 0, io_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out_r = __builtin_amdgcn_make_buffer_rsrc(A.out + s * A.out_stream_stride + A.out_ch_stride, 0,
                                                                             io_bytes, 0x00020000);
    for (int i = i0; i < i1; ++i) {
        {
            xb_v4f zp[8];
            forward(i, zp);
            products(zp, acc, carry, false, i + 1 >= i1);
        }
        // ---- XP -> sixteen 1024-point inverse transforms -> X1 -> radix-16, second half only -> out ------------------
        __syncthreads();                                // every thread has read its pairs of Z
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            lds_st(&X[pbase + 1040 * p], make_float2(acc[p][0].x, acc[p][0].y));
            lds_st(&X[kXbPlane + pbase + 1040 * p], make_float2(acc[p][1].x, acc[p][1].y));
        }
        __syncthreads();
        float2 v[16];
#pragma unroll
        for (int R = 0; R < 16; ++R) v[R] = lds_ld(&X[xb_pair_addr(64 * lane + 16 * (R & 3) + 4096 * (R >> 2) + c)]);
        __syncthreads();                                // every wave has its bins: the slabs are transform scratch again
        if (!(dskip & 16)) wave_fft_inv(v, slab, tab, lane);               // register a holds y_c[64 a + lane]
#pragma unroll
        for (int a = 0; a < 16; ++a) lds_st(&slab[64 * a + lane], v[a]);
        __syncthreads();
        float2 y[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) y[cc] = lds_ld(&X[kXbSlab * cc + tid]);
#pragma unroll
        for (int cc = 1; cc < 16; ++cc) y[cc] = cmulc(y[cc], twiddle(cc));
        if (!(dskip & 64)) dft16<true>(y);
#pragma unroll
        for (int n2 = 8; n2 < 16; ++n2) {
            const long long f0 = (long long)i * 8192 + 1024 * (n2 - 8);
            if (f0 + half < A.io_frames && !(dskip & 32)) {     // (uniform over the wave)
                const int so = __builtin_amdgcn_readfirstlane((int)(f0 * 4));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[n2].x * A.gain), r_out_l, v4, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[n2].y * A.gain), r_out_r, v4, so, 0);
            }
        }
        // (the next forward() starts with a barrier: X1 has been read by then)
    }
}

// ---------------------------------------------------------------------------------
// Second form (ConvXbArgs::form == 2): 8 waves, wave w transforms the classes c = w AND 16 - w (wave 0: 0 and 8)
// ---------------------------------------------------------------------------------
// Bin k = 16 k1 + c has its mirror N - k = 16 (1023 - k1) + (16 - c) in the OTHER class of the same wave, at lane 63 - L and
// register 15 - R of the 1024-point transform's bin layout (class 0 mirrors into itself by (1024 - k1) & 1023: wave_mirror;
// class 8 into itself by the reversal): the packed two-channel product W[k] = Z[k] C[k] + conj(Z[N - k]) D[k] finds its second
// input by a lane reversal (ds_bpermute) inside the wave -- no pair image, no workgroup barrier between the forward transform
// and the inverse one, and the tables are stored per bin in the transforms' own layout ([partition][wave][class][register][lane]).
// 512 threads (the samples n1 = 2 tid, 2 tid + 1 per thread in the radix-16 steps: 8-byte global and 16-byte LDS accesses).
__device__ __forceinline__ float2 xc_rev(float2 v, int src)
{
    float2 o;
    o.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.x)));
    o.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.y)));
    return o;
}

template <int P2>
__global__ __launch_bounds__(512) void k_conv_xc(const ConvXbArgs A)
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
    const int dskip = A.debug_skip;
    auto twiddle2 = [&](int cc, float2 &t0, float2 &t1) {      // w16384^((2 tid + e) cc), cc = 1 .. 15
        if (dskip & 2) { t0 = make_float2(1.0f, 0.001f * cc); t1 = t0; return; }
        const xb_v4f t = __builtin_bit_cast(xb_v4f, __builtin_amdgcn_raw_buffer_load_b128(r_tw, v16, (cc - 1) * 8192, 0));
        t0 = make_float2(t.x, t.y); t1 = make_float2(t.z, t.w);
    };
    // one sixteenth (1024 frames from frame f0 of the call's axis) of a window: the thread's frames f0 + 2 tid, + 1
    auto load16th = [&](long long f0, xb_v2f &l, xb_v2f &r) {
        l = (xb_v2f){0.f, 0.f}; r = (xb_v2f){0.f, 0.f};
        if (dskip & 1) return;
        if (f0 >= 0) {
            if (f0 + half < A.io_frames) {              // (uniform over the wave: everything is a multiple of 512)
                const int so = __builtin_amdgcn_readfirstlane((int)(f0 * 4));
                l = __builtin_bit_cast(xb_v2f, __builtin_amdgcn_raw_buffer_load_b64(r_in_l, v8, so, 0));
                r = __builtin_bit_cast(xb_v2f, __builtin_amdgcn_raw_buffer_load_b64(r_in_r, v8, so, 0));
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
    auto forward = [&](int i, float2 (&ZA)[16], float2 (&ZB)[16]) {
        const long long w0 = A.tables_mode ? (long long)i * 8192 : ((long long)i - 1) * 8192;
        float2 z0[16], z1[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) {
            xb_v2f l = {0.f, 0.f}, r = {0.f, 0.f};
            if (!A.tables_mode || n2 < 8) load16th(w0 + 1024 * n2, l, r);
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
    const __amdgpu_buffer_rsrc_t r_cd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(A.cd), 0, P2 * 2 * kXbPairs * 16, 0x00020000);
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
            forward(i, ZA, ZB);
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
        forward(i0 - 1, ZA, ZB);
        products(ZA, ZB, WA, WB, CA, CB, true, false);
    }
    const __amdgpu_buffer_rsrc_t r_out_l = __builtin_amdgcn_make_buffer_rsrc(A.out + s * A.out_stream_stride, 0, io_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out_r = __builtin_amdgcn_make_buffer_rsrc(A.out + s * A.out_stream_stride + A.out_ch_stride, 0,
                                                                             io_bytes, 0x00020000);
    for (int i = i0; i < i1; ++i) {
        {
            float2 ZA[16], ZB[16];
            forward(i, ZA, ZB);
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
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xb_v2u, ol), r_out_l, v8, so, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xb_v2u, orr), r_out_r, v8, so, 0);
            }
        }
        // (the next forward() starts with a barrier: X1 has been read by then)
    }
}

// form 2's tables: per bin, in the transforms' own layout; ab holds the spectra of A and B as float2 per bin
__global__ __launch_bounds__(256) void k_conv_xc_build_cd(const float2 *__restrict__ ab, int P2, float4 *__restrict__ cd)
{
    const int i = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;      // q < 16384
    const float2 a = ab[((size_t)0 * P2 + i) * kXbFft + q];
    const float2 b = ab[((size_t)1 * P2 + i) * kXbFft + q];
    const float h = 0.5f / (float)kXbFft;
    cd[(size_t)i * kXbFft + q] = make_float4(h * (a.x + b.y), h * (a.y - b.x), h * (a.x - b.y), h * (a.y + b.x));
}

// (C, D) tables from the spectra of A = h_lsl + j h_lsr (ab[0][i]) and B = h_rsl + j h_rsr (ab[1][i]), i < P2
__global__ __launch_bounds__(256) void k_conv_xb_build_cd(const float4 *__restrict__ ab, int P2, float4 *__restrict__ cd)
{
    const int i = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    const float4 a = ab[((size_t)0 * P2 + i) * kXbPairs + q];
    const float4 b = ab[((size_t)1 * P2 + i) * kXbPairs + q];
    const float h = 0.5f / (float)kXbFft;       // the 1/2 of C, D and the 1/N of the inverse transform (powers of two)
    cd[((size_t)i * 2 + 0) * kXbPairs + q] = make_float4(h * (a.x + b.y), h * (a.y - b.x), h * (a.x - b.y), h * (a.y + b.x));
    cd[((size_t)i * 2 + 1) * kXbPairs + q] = make_float4(h * (a.z + b.w), h * (a.w - b.z), h * (a.z - b.w), h * (a.w + b.z));
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
    for (int run = 1; run <= n_blk; ++run) {
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
    if (a.io_frames < 0 || (a.io_frames & 511) || (a.xhist && (a.xh_len <= 0 || (a.xh_len & 511) || a.xh_head < 0 || a.xh_head >= 2 * a.xh_len)))
        return hipErrorInvalidValue;
    static std::atomic<unsigned long long> ok1{0}, ok2{0}, ok3{0}, ok4{0};
    const int runs = (a.n_blk + a.run - 1) / a.run;
    const unsigned grid = xb_grid(runs, a.n_streams);
    hipError_t e;
    if (a.form == 2) {
        if (a.P2 == 2 && !a.tables_mode) {
            e = xb_allow_lds(reinterpret_cast<const void *>(&k_conv_xc<2>), ok3);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_conv_xc<2>, dim3(grid), dim3(512), kXbLdsBytes, st, a);
        } else {
            e = xb_allow_lds(reinterpret_cast<const void *>(&k_conv_xc<1>), ok4);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_conv_xc<1>, dim3(grid), dim3(512), kXbLdsBytes, st, a);
        }
        return hipGetLastError();
    }
    if (a.P2 == 2 && !a.tables_mode) {
        e = xb_allow_lds(reinterpret_cast<const void *>(&k_conv_xb<2>), ok2);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_conv_xb<2>, dim3(grid), dim3(1024), kXbLdsBytes, st, a);
    } else {
        e = xb_allow_lds(reinterpret_cast<const void *>(&k_conv_xb<1>), ok1);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_conv_xb<1>, dim3(grid), dim3(1024), kXbLdsBytes, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_conv_xb_build_cd(const float4 *ab, int P2, float4 *cd, hipStream_t st, int form)
{
    if (!ab || !cd || P2 < 1 || P2 > 2) return hipErrorInvalidValue;
    if (form == 2) hipLaunchKernelGGL(k_conv_xc_build_cd, dim3(kXbFft / 256, P2), dim3(256), 0, st, reinterpret_cast<const float2 *>(ab), P2, cd);
    else hipLaunchKernelGGL(k_conv_xb_build_cd, dim3(kXbPairs / 256, P2), dim3(256), 0, st, ab, P2, cd);
    return hipGetLastError();
}

}  // namespace ohs
