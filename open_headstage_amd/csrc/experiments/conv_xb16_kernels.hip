// conv_xb16_kernels.hip -- EXPERIMENTS BUILD ONLY: the first form of the block-8192 / FFT-16384 single-kernel plan (the product's is
// conv_xb_kernels.hip): SIXTEEN waves, one class each, spectra exchanged into PAIRS (slot q = (Z[q], Z[16384 - q])) through a second
// LDS image.  Parity-green (tests/test_gpu_conv_xb.py runs both forms), 0.152-0.165 ms per step on BASELINE config 4 against
// 0.133-0.145 for block 2048: 144.5 KiB of LDS mean one workgroup per CU, whose sixteen waves walk through the block's phases --
// load + radix-16, X1, 1024-point transforms, XP, products, XP, transforms, X1, radix-16 + store -- between EIGHT workgroup barriers,
// so vector unit (39 % busy), LDS and memory take turns: 33 us per block (profiles/r05_xb_phase_knockouts.txt), and the 128-register
// budget of sixteen waves leaves nothing to request ahead into.  Kept as the measured step it was; Tuning::xb_form = 1 selects it.
//   X1   class c, sample n1  at  1088 c + n1                                   thread (n1 = tid) <-> wave c (n1 = 64 a + lane)
//   XP   bin k               at  8704 (k >= 8192) + 65 (q >> 6) + (q & 63),  q = k < 8192 ? k : (16384 - k) & 8191
//                                                                               wave c (k = 64 lane + ..) <-> thread (q = tid + 1024 p)
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

// workgroup -> (stream, run) with all runs of a stream on ONE XCD (conv_lb_kernels.hip: lb_job)
__device__ __forceinline__ bool xb16_job(int per_stream, int n_streams, size_t &s, int &idx)
{
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    s = (size_t)(slot / (unsigned)per_stream) * 8 + xcd;
    idx = (int)(slot % (unsigned)per_stream);
    return s < (size_t)n_streams;
}
static unsigned xb16_grid(int per_stream, int n_streams) { return (unsigned)(((n_streams + 7) / 8) * 8 * per_stream); }

// acc += z c + conj(m) d on (re, im) register pairs (conv_lb_kernels.hip: lb_cmac)
__device__ __forceinline__ void xb16_cmac(xb_v2f &acc, xb_v2f z, xb_v2f c, xb_v2f m, xb_v2f d)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(z), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(acc) : "v"(z), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(m), "v"(d));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "+v"(acc) : "v"(m), "v"(d));
}

// P2: partitions of 8192 taps (1 or 2).  One workgroup = 16 waves = (stream, run of A.run consecutive blocks).
template <int P2>
__global__ __launch_bounds__(1024) void k_conv_xb16(const ConvXbArgs A)
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
    if (!xb16_job(runs, A.n_streams, s, chunk)) return;
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
                    xb16_cmac(acc[p][0], zz.lo, t[e][0][0].lo, m0, t[e][0][0].hi);
                    xb16_cmac(acc[p][1], zz.hi, t[e][0][1].lo, m1, t[e][0][1].hi);
                }
                if constexpr (P2 == 2) {
                    carry[p][0] = (xb_v2f){0.f, 0.f}; carry[p][1] = (xb_v2f){0.f, 0.f};
                    if (!last) {
                        xb16_cmac(carry[p][0], zz.lo, t[e][1][0].lo, m0, t[e][1][0].hi);
                        xb16_cmac(carry[p][1], zz.hi, t[e][1][1].lo, m1, t[e][1][1].hi);
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

// (C, D) tables from the spectra of A = h_lsl + j h_lsr (ab[0][i]) and B = h_rsl + j h_rsr (ab[1][i]), i < P2
__global__ __launch_bounds__(256) void k_conv_xb16_build_cd(const float4 *__restrict__ ab, int P2, float4 *__restrict__ cd)
{
    const int i = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    const float4 a = ab[((size_t)0 * P2 + i) * kXbPairs + q];
    const float4 b = ab[((size_t)1 * P2 + i) * kXbPairs + q];
    const float h = 0.5f / (float)kXbFft;       // the 1/2 of C, D and the 1/N of the inverse transform (powers of two)
    cd[((size_t)i * 2 + 0) * kXbPairs + q] = make_float4(h * (a.x + b.y), h * (a.y - b.x), h * (a.x - b.y), h * (a.y + b.x));
    cd[((size_t)i * 2 + 1) * kXbPairs + q] = make_float4(h * (a.z + b.w), h * (a.w - b.z), h * (a.z - b.w), h * (a.w + b.z));
}

static hipError_t xb16_allow_lds(const void *fn, std::atomic<unsigned long long> &done_mask)
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
hipError_t launch_conv_xb16(const ConvXbArgs &a, hipStream_t st)
{
    if (a.n_streams <= 0 || a.n_blk <= 0 || a.run <= 0 || !a.in || !a.tw || !a.tw16384 || (a.P2 != 1 && a.P2 != 2))
        return hipErrorInvalidValue;
    if (a.tables_mode ? !a.ab : (!a.out || !a.cd)) return hipErrorInvalidValue;
    if (a.io_frames < 0 || (a.io_frames & 511) || (a.xhist && (a.xh_len <= 0 || (a.xh_len & 511) || a.xh_head < 0 || a.xh_head >= 2 * a.xh_len)))
        return hipErrorInvalidValue;
    static std::atomic<unsigned long long> ok1{0}, ok2{0};
    const int runs = (a.n_blk + a.run - 1) / a.run;
    const unsigned grid = xb16_grid(runs, a.n_streams);
    hipError_t e;
    if (a.P2 == 2 && !a.tables_mode) {
        e = xb16_allow_lds(reinterpret_cast<const void *>(&k_conv_xb16<2>), ok2);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_conv_xb16<2>, dim3(grid), dim3(1024), kXbLdsBytes, st, a);
    } else {
        e = xb16_allow_lds(reinterpret_cast<const void *>(&k_conv_xb16<1>), ok1);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_conv_xb16<1>, dim3(grid), dim3(1024), kXbLdsBytes, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_conv_xb16_build_cd(const float4 *ab, int P2, float4 *cd, hipStream_t st)
{
    if (!ab || !cd || P2 < 1 || P2 > 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv_xb16_build_cd, dim3(kXbPairs / 256, P2), dim3(256), 0, st, ab, P2, cd);
    return hipGetLastError();
}

}  // namespace ohs
