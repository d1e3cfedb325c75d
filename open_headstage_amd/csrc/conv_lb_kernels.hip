// conv_lb_kernels.hip -- long impulse responses in batch mode: block 2048 / FFT 4096, overlap-save ("large block" plan).
//
// What it replaces: the reference's uniform blocking (src/dsp/convolution.rs:22-23: block 512 / FFT 1024; :120-132: one
// spectrum per 512 taps; :261-273: P x 1024 complex products per block).  BASELINE config 4 (16 384 taps, P = 32) spends
// 61 % of its time in that product and moves 6.4 x its I/O floor through HBM (history ring written and read, product
// written and read).  With 2048-tap partitions the products and the history reads per frame drop to a quarter, and with
// the product and the inverse transform in ONE kernel the product never travels through HBM:
//   per frame (stereo): audio in 8 B + ring write 16 B | ring read 16 B + audio out 8 B   (block 512: 8 + 16 | 16 + 16 | 16 + 8)
//
// The 4096-point transform = four of the wave's 1024-point transforms (wave_fft.hpp), one per wave of a 4-wave
// workgroup, plus a lane-local radix-4 step (tools/proto_lb_conv.py is the index-exact CPU model):
//   forward, decimation in frequency:  v_c[n1] = w4096^(n1 c) sum_n2 w4^(n2 c) z[n1 + 1024 n2];  Z[4 k1 + c] = FFT1024(v_c)[k1]
//   inverse, decimation in time:       y_c = IFFT1024(W[4 k1 + c]);  x[n1 + 1024 n2] = sum_c w4^(-n2 c) conj(w4096^(n1 c)) y_c[n1]
// Overlap-save: window u = frames [(u - 1) 2048, (u + 1) 2048), output block t = the second half of IFFT(W_t) -- only
// n2 = 2, 3 of the last radix-4 step are computed, and no overlap state exists: the state of a stream at rest is its
// last P2 x 2048 INPUT frames (ConvState::d_xhist, a ring), from which the block-512 state is rebuilt when another plan needs it.
//
// Spectra are stored as PAIRS: slot q = (Z[q], Z[4096 - q]), slot 0 = (Z[0], Z[2048]) -- the packed two-channel product
//   W[k] = Z[k] C[k] + conj(Z[N - k]) D[k]    (C = (A - jB)/2, D = (A + jB)/2, A = H_lsl + j H_lsr, B = H_rsl + j H_rsr)
// then finds both of its inputs in one 16-byte load, and every global access is lane-contiguous.
//
// LDS of a workgroup: the 8.5 KiB twiddle tables + four slabs of 1088 complex values (42.5 KiB in all).  The slabs are the
// four waves' transform scratch AND the three exchange images (never at the same time; workgroup barriers in between):
//   X14  class c, sample n1      at  1088 c + 264 (n1 & 3) + (n1 >> 2)      thread (n1 = 4 tid + e) <-> wave c (n1 = 64 a + lane)
//   X23  bin k                   at  2176 (k >= 2048) + 17 (q >> 4) + (q & 15),  q = k < 2048 ? k : (4096 - k) & 2047
//                                                                            wave c (k = 16 lane + ..) <-> thread (q = tid + 256 p)
// Both are conflict-free for the strided side (lane stride 17 complex = 34 dwords; 264 = 8 mod 32 complex) and contiguous
// for the other.
#include "kernels.h"
#include "wave_fft.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>

namespace ohs {

typedef float lb_v2f __attribute__((ext_vector_type(2)));
typedef float lb_v4f __attribute__((ext_vector_type(4)));
typedef unsigned lb_v4u __attribute__((ext_vector_type(4)));

constexpr int kLbFft = 4096;
constexpr int kLbPairs = 2048;
constexpr int kLbSlab = 1088;                   // complex values per wave slab (the 1024-point transform needs 1084)
constexpr int kLbPlane = 2176;                  // X23: complex values per component plane = 128 rows x 17
constexpr size_t kLbLdsBytes = (kTabComplex + 4 * (size_t)kLbSlab) * sizeof(float2);
static_assert(kLbSlab >= kWaveLdsComplex && 2 * kLbSlab == kLbPlane && 3 * 264 + 255 < kLbSlab, "LDS plan");

__device__ __forceinline__ int lb_x14(int c, int n1) { return kLbSlab * c + 264 * (n1 & 3) + (n1 >> 2); }
__device__ __forceinline__ int lb_x23(int k)
{
    const int hi = k >= kLbPairs;
    const int q = hi ? ((kLbFft - k) & (kLbPairs - 1)) : k;
    return kLbPlane * hi + 17 * (q >> 4) + (q & 15);
}

// host: w4096^(n1 c), c = 1 .. 3, from f64 (as the 1024-point tables are)
void conv_lb_build_twiddles(float2 *out)
{
    for (int c = 1; c < 4; ++c)
        for (int n1 = 0; n1 < 1024; ++n1) {
            const double ang = -2.0 * 3.14159265358979323846264338327950288 * (double)(n1 * c) / 4096.0;
            out[(c - 1) * 1024 + n1] = make_float2((float)cos(ang), (float)sin(ang));
        }
}

// workgroup -> (stream, index) with all workgroups of a stream on ONE XCD (the dispatcher deals workgroups to the 8 XCDs
// round-robin by linear id; a stream's windows / tiles share input blocks and ring slots through that XCD's L2)
__device__ __forceinline__ bool lb_job(int per_stream, int n_streams, size_t &s, int &idx)
{
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    s = (size_t)(slot / (unsigned)per_stream) * 8 + xcd;
    idx = (int)(slot % (unsigned)per_stream);
    return s < (size_t)n_streams;
}
static unsigned lb_grid(int per_stream, int n_streams) { return (unsigned)(((n_streams + 7) / 8) * 8 * per_stream); }

// ---------------------------------------------------------------------------------
// F: one workgroup (4 waves) per (stream, run of A.run consecutive windows)
// ---------------------------------------------------------------------------------
// A window is [block u - 1, block u]: the workgroup keeps the block it has just transformed as the next window's first half
// (every input frame is loaded once per run, not twice), and requests the block after that before it starts on the
// current window -- the loads' latency hides behind a whole transform instead of standing in front of one.
struct LbBlock { lb_v4f l[2], r[2]; };          // one 2048-frame block: quarter q, frames 1024 q + 4 tid .. + 3, both channels

__global__ __launch_bounds__(256) void k_conv_lb_forward(const ConvLbArgs A)
{
    ohs_set_fp_mode(A.fp_mode);
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    float2 *X = smem + kTabComplex;
    const int tid = threadIdx.x;
    fill_twiddle_tables(tab, A.tw, tid, 256);
    const int runs = (A.n_fwd + A.run - 1) / A.run;
    size_t s;
    int chunk;
    if (!lb_job(runs, A.n_streams, s, chunk)) return;
    const int w0 = chunk * A.run, w1 = w0 + A.run < A.n_fwd ? w0 + A.run : A.n_fwd;
    const int c = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    const float *in_l = A.in + s * A.in_stream_stride, *in_r = in_l + A.in_ch_stride;
    const long long L2 = 2 * A.xh_len;              // the input history is a ring of twice its reach (kernels.h)
    const float *xh_l = A.xhist ? A.xhist + s * 2 * L2 : nullptr, *xh_r = xh_l ? xh_l + L2 : nullptr;
    float *xa_l = A.xh_append ? A.xh_append + s * 2 * L2 : nullptr, *xa_r = xa_l ? xa_l + L2 : nullptr;
    // block b of the grid (frames [2048 b, 2048 b + 2048)): from `in` (frames [io_frame0, io_frame0 + io_frames)), from the
    // input history in front of that, or zeros behind it.  own: the block is loaded as the second half of ITS window --
    // exactly once per launch -- and the buffer's last xh_cnt frames join the input history then, in the half of the ring
    // no window of this launch reads (they reach back xh_len at most).  (The source choice is wave-uniform: a quarter of
    // 1024 frames cut at 4 tid never straddles the buffer's start or end, both multiples of 512.)
    auto load_block = [&](long long b, bool own, LbBlock &o) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const long long f0 = b * kLbBlock + 1024 * q + 4 * tid - A.io_frame0;      // relative to the buffer's first frame
            lb_v4f l = {0.f, 0.f, 0.f, 0.f}, r = {0.f, 0.f, 0.f, 0.f};
            if (f0 >= 0) {
                if (f0 < A.io_frames) {
                    l = *reinterpret_cast<const lb_v4f *>(in_l + f0);
                    r = *reinterpret_cast<const lb_v4f *>(in_r + f0);
                    const long long fa = f0 - (A.io_frames - A.xh_cnt);
                    if (own && xa_l && fa >= 0) {
                        const long long pos = (A.xh_head + fa) % L2;
                        *reinterpret_cast<lb_v4f *>(xa_l + pos) = l;
                        *reinterpret_cast<lb_v4f *>(xa_r + pos) = r;
                    }
                }
            } else if (xh_l && A.xh_len + f0 >= 0) {
                long long idx = A.xh_head + f0;             // (head and the ring's length are multiples of 512: a
                idx = idx < 0 ? idx + L2 : idx;             //  4-frame group does not straddle the wrap)
                l = *reinterpret_cast<const lb_v4f *>(xh_l + idx);
                r = *reinterpret_cast<const lb_v4f *>(xh_r + idx);
            }
            o.l[q] = l; o.r[q] = r;
        }
    };
    const LbBlock zero = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
    // tables mode: window i = [block i, 2048 zeros]; else window u = [block u - 1, block u]
    LbBlock first, second, next = zero;
    if (A.tables_mode) { load_block(A.u0 + w0, false, first); second = zero; }
    else { load_block((long long)A.u0 + w0 - 1, false, first); load_block((long long)A.u0 + w0, true, second); }
    // w4096^(n1 c) for the thread's four samples n1 = 4 tid + e, c = 1 .. 3
    lb_v4f tw[3][2];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        const lb_v4f *t4 = reinterpret_cast<const lb_v4f *>(A.tw4096 + cc * 1024) + 2 * tid;
        tw[cc][0] = t4[0]; tw[cc][1] = t4[1];
    }
    float2 *slab = X + kLbSlab * c;
    const int xb = 17 * (tid >> 4) + (tid & 15);                    // lb_x23(tid + 256 p) = xb + 272 p (+ plane)

    for (int w = w0; w < w1; ++w) {
        const int u = A.u0 + w;
        if (w + 1 < w1) load_block((long long)u + 1, !A.tables_mode, next);       // the block the next window brings in
        // z[n2][e] = window sample 4 tid + e + 1024 n2
        float2 z[4][4];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z[q][e] = make_float2(first.l[q][e], first.r[q][e]);
                z[2 + q][e] = make_float2(second.l[q][e], second.r[q][e]);
            }
        // radix-4 over the quarters, then w4096^(n1 c): class c of sample n1 = 4 tid + e
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dft4<false>(z[0][e], z[1][e], z[2][e], z[3][e]);
#pragma unroll
            for (int cc = 1; cc < 4; ++cc) {
                const lb_v4f t = tw[cc - 1][e >> 1];
                const float2 wv = (e & 1) ? make_float2(t.z, t.w) : make_float2(t.x, t.y);
                z[cc][e] = cmul(z[cc][e], wv);
            }
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e) lds_st(&X[kLbSlab * cc + 264 * e + tid], z[cc][e]);      // X14: lb_x14(cc, 4 tid + e)
        __syncthreads();

        // wave c: the 1024-point transform of class c (time layout: lane m, register a holds sample 64 a + m)
        float2 v[16];
        {
            const int base = 264 * (lane & 3) + (lane >> 2);            // lb_x14(c, 64 a + lane) = slab + base + 16 a
#pragma unroll
            for (int a = 0; a < 16; ++a) v[a] = lds_ld(&slab[base + 16 * a]);
            wave_lds_sync();
        }
        wave_fft_fwd(v, slab, tab, lane);           // bin layout: register 4 r + j holds k1 = 4 lane + j + 256 r, i.e. bin 4 k1 + c
        __syncthreads();                            // every wave is done with its slab: the slabs become the X23 image
#pragma unroll
        for (int R = 0; R < 16; ++R) {
            const int k = 16 * lane + 4 * (R & 3) + c + 1024 * (R >> 2);
            lds_st(&X[lb_x23(k)], v[R]);
        }
        __syncthreads();
        // pairs out: slot q = tid + 256 p
        const long long sl = (A.ring_base + u) % A.ring_cap;
        float4 *dst = A.ring + (s * (size_t)A.ring_cap + (size_t)sl) * kLbPairs;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float2 a = lds_ld(&X[xb + 272 * p]), b = lds_ld(&X[kLbPlane + xb + 272 * p]);
            dst[tid + 256 * p] = make_float4(a.x, a.y, b.x, b.y);
        }
        if (w + 1 < w1) {
            __syncthreads();                        // the X23 image has been read: the next window's X14 image may be written
            if (A.tables_mode) first = next;
            else { first = second; second = next; }
        }
    }
}

// ---------------------------------------------------------------------------------
// MI: one workgroup per (stream, tile of kLbTile output blocks): pair products in registers, then the inverse transforms
// ---------------------------------------------------------------------------------
// acc += z c + conj(m) d on (re, im) register pairs: four v_pk_fma_f32 (conv_mac_kernels.hip explains the modifiers)
__device__ __forceinline__ void lb_cmac(lb_v2f &acc, lb_v2f z, lb_v2f c, lb_v2f m, lb_v2f d)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(z), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(acc) : "v"(z), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(m), "v"(d));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "+v"(acc) : "v"(m), "v"(d));
}

// PI: partitions per register window, 8 (P2pad a multiple of 8) or 4.  TB: output blocks per workgroup -- 4 for calls of
// several blocks (11 window spectra + 16 table rows per pair group feed 4 x 8 x 2 products: the tile that keeps the loads per
// product low), 2 or 1 for the short calls of a host that feeds the plugin's 1024-frame block (a 4-block tile would compute
// three blocks for nothing, and with one block's 32 accumulator registers the next pair group's operands can be in flight
// while this one's products run: a lone workgroup per CU is a chain of load latencies otherwise).
template <int PI, int TB>
__global__ __launch_bounds__(256, 2) void k_conv_lb_mac_inverse(const ConvLbArgs A)
{
    ohs_set_fp_mode(A.fp_mode);
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    float2 *X = smem + kTabComplex;
    const int tid = threadIdx.x;
    fill_twiddle_tables(tab, A.tw, tid, 256);
    const int tiles = (A.n_big + TB - 1) / TB;
    size_t s;
    int tile;
    if (!lb_job(tiles, A.n_streams, s, tile)) return;
    const int tb0 = tile * TB;
    const int c = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    // ---- pair products: thread = 8 pairs (q = tid + 256 p) x TB output blocks -------------------------------------
    lb_v2f acc[8][TB][2];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int t = 0; t < TB; ++t) { acc[p][t][0] = (lb_v2f){0.f, 0.f}; acc[p][t][1] = (lb_v2f){0.f, 0.f}; }
    // Operands come through buffer loads: one resource for the stream's ring, one for the tables of the partition window,
    // the thread's 16 bytes at voffset = 16 tid, everything else -- ring slot, pair group p, partition -- in the SCALAR
    // offset.  (With flat pointers hipcc kept a 64-bit address pair per window slot and table row alive: ~50 VGPRs, and
    // with 128 accumulators that left it two loads in flight at a time.)
    const int voff = tid * 16;
    const __amdgpu_buffer_rsrc_t r_ring = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float4 *>(A.ring) + s * (size_t)A.ring_cap * kLbPairs, 0, A.ring_cap * (kLbPairs * 16), 0x00020000);
    const bool special = tid == 0;              // pair slot 0 holds bins 0 and 2048, each its own mirror
#ifdef OHS_EXPERIMENTS
    const int dskip = A.debug_skip;
#else
    constexpr int dskip = 0;
#endif
    for (int i0 = 0; i0 < ((dskip & 1) ? 0 : A.P2pad); i0 += PI) {
        const __amdgpu_buffer_rsrc_t r_cd = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float4 *>(A.cd) + (size_t)i0 * 2 * kLbPairs, 0, PI * 2 * (kLbPairs * 16), 0x00020000);
        // window of blocks t - i, t in [0, TB), i in [i0, i0 + PI): w = t - (i - i0) + PI - 1 in [0, PI + TB - 1)
        int so[PI + TB - 1];
        {
            int sl = (int)((A.ring_base + tb0 - i0 - (PI - 1)) % A.ring_cap);
#pragma unroll
            for (int w = 0; w < PI + TB - 1; ++w) {
                so[w] = __builtin_amdgcn_readfirstlane(sl * (kLbPairs * 16));
                sl = sl + 1 == A.ring_cap ? 0 : sl + 1;
            }
        }
        // one pair group's operands: W window spectra + 2 x PI table rows, all requested at once
        constexpr int W = PI + TB - 1;
        struct Ops { lb_v4f zw[W], cc[2][PI]; };
        auto request = [&](int p, Ops &o) {
#pragma unroll
            for (int w = 0; w < W; ++w)
                o.zw[w] = __builtin_bit_cast(lb_v4f, __builtin_amdgcn_raw_buffer_load_b128(r_ring, (dskip & 4) ? 0x7ffffff0 : voff,
                                                                                           so[w] + p * 4096, 0));
#pragma unroll
            for (int comp = 0; comp < 2; ++comp)
#pragma unroll
                for (int ii = 0; ii < PI; ++ii)
                    o.cc[comp][ii] = __builtin_bit_cast(lb_v4f, __builtin_amdgcn_raw_buffer_load_b128(
                                                                    r_cd, (dskip & 8) ? 0x7ffffff0 : voff,
                                                                    (ii * 2 + comp) * (kLbPairs * 16) + p * 4096, 0));
        };
        auto products = [&](int p, const Ops &o) {
#pragma unroll
            for (int comp = 0; comp < 2; ++comp)
#pragma unroll
                for (int ii = 0; ii < PI; ++ii)
#pragma unroll
                    for (int t = 0; t < TB; ++t) {
                        const lb_v4f zz = o.zw[t - ii + PI - 1];
                        lb_v2f z = comp ? zz.hi : zz.lo, m = comp ? zz.lo : zz.hi;
                        if (p == 0) m = special ? z : m;
                        lb_cmac(acc[p][t][comp], z, o.cc[comp][ii].lo, m, o.cc[comp][ii].hi);
                    }
        };
        if constexpr (TB == 1) {
            // two operand sets: group p + 1 is in flight while group p's products run
            Ops oa, ob;
            request(0, oa);
#pragma unroll
            for (int p = 0; p < 8; p += 2) {
                request(p + 1, ob);
                __builtin_amdgcn_sched_barrier(0);
                products(p, oa);
                __builtin_amdgcn_sched_barrier(0);
                if (p + 2 < 8) request(p + 2, oa);
                __builtin_amdgcn_sched_barrier(0);
                products(p + 1, ob);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                Ops o;
                request(p, o);
                __builtin_amdgcn_sched_barrier(0);
                products(p, o);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();            // (the twiddle tables are in place)

#ifdef OHS_EXPERIMENTS
    if (dskip & 2) {    // (keep the products alive: one sum per thread)
        lb_v2f sum = {0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int t = 0; t < TB; ++t) sum += acc[p][t][0] + acc[p][t][1];
        if (sum.x == 12345.678f) A.out[s * A.out_stream_stride + tid] = sum.y;
    }
#endif
    // ---- per output block: X23 -> four 1024-point inverse transforms -> X14 -> radix-4, second half only -> out --------
    float2 *slab = X + kLbSlab * c;
    const int xb = 17 * (tid >> 4) + (tid & 15);
    float *out_l = A.out + s * A.out_stream_stride, *out_r = out_l + A.out_ch_stride;
    lb_v4f tw[3][2];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        const lb_v4f *t4 = reinterpret_cast<const lb_v4f *>(A.tw4096 + cc * 1024) + 2 * tid;
        tw[cc][0] = t4[0]; tw[cc][1] = t4[1];
    }
#pragma unroll
    for (int t = 0; t < TB; ++t) {
        if (tb0 + t >= A.n_big || (dskip & 2)) break;          // (uniform over the workgroup)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            lds_st(&X[xb + 272 * p], make_float2(acc[p][t][0].x, acc[p][t][0].y));
            lds_st(&X[kLbPlane + xb + 272 * p], make_float2(acc[p][t][1].x, acc[p][t][1].y));
        }
        __syncthreads();
        float2 v[16];
#pragma unroll
        for (int R = 0; R < 16; ++R) {
            const int k = 16 * lane + 4 * (R & 3) + c + 1024 * (R >> 2);
            v[R] = lds_ld(&X[lb_x23(k)]);
        }
        __syncthreads();        // every wave has its bins: the slabs are transform scratch again
        wave_fft_inv(v, slab, tab, lane);       // time layout: register a holds y_c[64 a + lane]
        {
            // X14 with a row pitch of 260 here (the forward kernel's is 264): this side WRITES with the strided pattern, and
            // ds_write_b64 goes by 16 lanes over 32 banks -- 2 x 260 = 8 mod 32 spreads a group's four rows, 2 x 264 = 16 does not
            const int base = 260 * (lane & 3) + (lane >> 2);
#pragma unroll
            for (int a = 0; a < 16; ++a) lds_st(&slab[base + 16 * a], v[a]);
        }
        __syncthreads();
        lb_v4f ol[2], orr[2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float2 y[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) y[cc] = lds_ld(&X[kLbSlab * cc + 260 * e + tid]);
#pragma unroll
            for (int cc = 1; cc < 4; ++cc) {
                const lb_v4f tt = tw[cc - 1][e >> 1];
                const float2 wv = (e & 1) ? make_float2(tt.z, tt.w) : make_float2(tt.x, tt.y);
                y[cc] = cmulc(y[cc], wv);
            }
            // n2 = 2: y0 - y1 + y2 - y3 ; n2 = 3: y0 - j y1 - y2 + j y3
            const float2 s02 = cadd(y[0], y[2]), d02 = csub(y[0], y[2]), s13 = cadd(y[1], y[3]), d13 = csub(y[1], y[3]);
            const float2 x2 = csub(s02, s13);
            const float2 x3 = make_float2(d02.x + d13.y, d02.y - d13.x);
            ol[0][e] = x2.x * A.gain; orr[0][e] = x2.y * A.gain;
            ol[1][e] = x3.x * A.gain; orr[1][e] = x3.y * A.gain;
        }
        const long long f0 = ((long long)tb0 + t) * kLbBlock + 4 * tid - A.io_frame0;      // relative to the buffer's first frame
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long f = f0 + 1024 * h;
            if (f >= 0 && f < A.io_frames) {
                *reinterpret_cast<lb_v4f *>(out_l + f) = ol[h];
                *reinterpret_cast<lb_v4f *>(out_r + f) = orr[h];
            }
        }
        __syncthreads();        // X14 has been read: the next block's X23 image may be written
    }
}

// ---------------------------------------------------------------------------------
// tables, state
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_lb_build_cd(const float4 *__restrict__ ab, int P2pad, float4 *__restrict__ cd, int mask)
{
    const int i = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    // mask bit 0: the L-speaker pair (A = H_lsl + j H_lsr) takes part, bit 1: the R-speaker pair (B = H_rsl + j H_rsr)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 a = (mask & 1) ? ab[((size_t)0 * P2pad + i) * kLbPairs + q] : z4;
    const float4 b = (mask & 2) ? ab[((size_t)1 * P2pad + i) * kLbPairs + q] : z4;
    const float h = 0.5f / (float)kLbFft;       // the 1/2 of C, D and the 1/N of the inverse transform (powers of two)
    // C = (A - jB)/2, D = (A + jB)/2 per component (k_build_cd_multi does the same for the 1024-point spectra)
    cd[((size_t)i * 2 + 0) * kLbPairs + q] = make_float4(h * (a.x + b.y), h * (a.y - b.x), h * (a.x - b.y), h * (a.y + b.x));
    cd[((size_t)i * 2 + 1) * kLbPairs + q] = make_float4(h * (a.z + b.w), h * (a.w - b.z), h * (a.z - b.w), h * (a.w + b.z));
}

__global__ __launch_bounds__(256) void k_conv_lb_xhist_append(const float *__restrict__ in, long long in_ss, long long in_cs,
                                                              float *__restrict__ ring, long long len, long long head,
                                                              long long n_frames, long long count, unsigned wgs_per_row)
{
    const long long j = 4 * ((long long)(blockIdx.x % wgs_per_row) * 256 + threadIdx.x);
    if (j >= count) return;
    const size_t row = blockIdx.x / wgs_per_row;    // stream * 2 + channel
    const long long pos = (head + j) % (2 * len);   // (count = min(n_frames, len); everything a multiple of 512)
    *reinterpret_cast<lb_v4f *>(ring + row * 2 * len + pos) =
        *reinterpret_cast<const lb_v4f *>(in + (row >> 1) * in_ss + (row & 1) * in_cs + (n_frames - count + j));
}

__global__ __launch_bounds__(256) void k_conv_lb_xhist_linearise(const float *__restrict__ ring, float *__restrict__ dst,
                                                                 long long len, long long head, unsigned wgs_per_row)
{
    const long long j = 4 * ((long long)(blockIdx.x % wgs_per_row) * 256 + threadIdx.x);
    if (j >= len) return;
    const size_t row = blockIdx.x / wgs_per_row;
    const long long pos = (head + len + j) % (2 * len);     // = head - len + j
    *reinterpret_cast<lb_v4f *>(dst + row * len + j) = *reinterpret_cast<const lb_v4f *>(ring + row * 2 * len + pos);
}

static hipError_t lb_allow_lds(const void *fn, std::atomic<unsigned long long> &done_mask)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done_mask.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLbLdsBytes);
    if (e == hipSuccess) done_mask.fetch_or(bit, std::memory_order_release);
    return e;
}

static bool lb_args_ok(const ConvLbArgs &a)
{
    if (a.n_streams <= 0 || a.ring_cap <= 0 || !a.ring || !a.tw || !a.tw4096) return false;
    if ((a.in_stream_stride | a.in_ch_stride | a.out_stream_stride | a.out_ch_stride | a.xh_len) & 3) return false;
    if ((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(a.xhist) |
         reinterpret_cast<uintptr_t>(a.xh_append)) & 15)
        return false;
    if (a.xhist && (a.xh_len <= 0 || (a.xh_len & 511) || (a.xh_head & 511) || a.xh_head < 0 || a.xh_head >= 2 * a.xh_len)) return false;
    if (a.xh_append && (a.xh_append != a.xhist || a.xh_cnt < 0 || a.xh_cnt > a.xh_len || a.xh_cnt > a.io_frames || (a.xh_cnt & 511))) return false;
    if (a.io_frames < 0 || (a.io_frames & 511) || a.io_frame0 < 0 || (a.io_frame0 & 511)) return false;
    return true;
}

hipError_t launch_conv_lb_forward(const ConvLbArgs &a, hipStream_t st)
{
    if (a.n_fwd <= 0) return hipSuccess;
    if (!lb_args_ok(a) || !a.in || a.n_fwd > a.ring_cap || a.ring_base + a.u0 < 0) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> lds_ok{0};
    const hipError_t e = lb_allow_lds(reinterpret_cast<const void *>(k_conv_lb_forward), lds_ok);
    if (e != hipSuccess) return e;
    // windows per workgroup: as few as fill the chip's workgroup slots ONCE (three per CU: 42.5 KiB of LDS each) -- a
    // run's first window loads one block more than the others, and a second round of workgroups would start cold
    ConvLbArgs f = a;
    {
        static std::atomic<int> cus_cached{0};
        int cus = cus_cached.load(std::memory_order_relaxed);
        if (cus == 0) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                cus = 256;
            cus_cached.store(cus, std::memory_order_relaxed);
        }
        const long long slots = 3ll * cus, windows = (long long)a.n_streams * a.n_fwd;
        long long run = (windows + slots - 1) / slots;
        run = std::max<long long>(1, std::min<long long>(run, 16));
        if (a.run > 0) run = a.run;             // (a caller's choice: tests)
        f.run = (int)std::min<long long>(run, a.n_fwd);
    }
    const int runs = (f.n_fwd + f.run - 1) / f.run;
    hipLaunchKernelGGL(k_conv_lb_forward, dim3(lb_grid(runs, f.n_streams)), dim3(256), kLbLdsBytes, st, f);
    return hipGetLastError();
}

hipError_t launch_conv_lb_mac_inverse(const ConvLbArgs &a, hipStream_t st, int *tile_blocks)
{
    if (a.n_big <= 0) return hipSuccess;
    // the oldest window a tile reads is block 0's partition P2pad - 1; a tile's unused blocks read slots that alias others
    if (!lb_args_ok(a) || !a.out || !a.cd || a.P2pad < 4 || (a.P2pad & 3) || a.n_big + a.P2pad - 1 > a.ring_cap ||
        a.ring_base - (a.P2pad - 1) - 8 < 0)
        return hipErrorInvalidValue;
    // output blocks per workgroup: 4, or as many as the call has (1 or 2: the short calls)
    const int TB = a.n_big >= 3 ? kLbTile : a.n_big;
    if (tile_blocks) *tile_blocks = TB;
    const int tiles = (a.n_big + TB - 1) / TB;
    const dim3 grid(lb_grid(tiles, a.n_streams));
#define OHS_LB_MI(PI_, TB_)                                                                                                  \
    do {                                                                                                                     \
        static std::atomic<unsigned long long> lds_ok{0};                                                                    \
        const hipError_t e = lb_allow_lds(reinterpret_cast<const void *>(k_conv_lb_mac_inverse<PI_, TB_>), lds_ok);          \
        if (e != hipSuccess) return e;                                                                                       \
        hipLaunchKernelGGL((k_conv_lb_mac_inverse<PI_, TB_>), grid, dim3(256), kLbLdsBytes, st, a);                          \
    } while (0)
    if (a.P2pad % 8 == 0) {
        if (TB == 1) OHS_LB_MI(8, 1); else if (TB == 2) OHS_LB_MI(8, 2); else OHS_LB_MI(8, 4);
    } else {
        if (TB == 1) OHS_LB_MI(4, 1); else if (TB == 2) OHS_LB_MI(4, 2); else OHS_LB_MI(4, 4);
    }
#undef OHS_LB_MI
    return hipGetLastError();
}

hipError_t launch_conv_lb_build_cd(const float4 *ab, int P2pad, float4 *cd, hipStream_t st, int mask)
{
    hipLaunchKernelGGL(k_conv_lb_build_cd, dim3(kLbPairs / 256, P2pad), dim3(256), 0, st, ab, P2pad, cd, mask);
    return hipGetLastError();
}

// ---- pending tails (a per-path set_ir in mid-stream; api_conv.hip: conv_lb_tail_route) ---------------------------------
// dst[s][q][j] = keep(q) ? pend[s][q][pos + j] + fresh[s][q & 1][j] : 0   for the two paths q = 2 pair, 2 pair + 1
__global__ __launch_bounds__(256) void k_conv_lb_tails_merge(const float *__restrict__ pend, long long pos, const float *__restrict__ fresh,
                                                             float *__restrict__ dst, long long len, int pair, int keep_mask,
                                                             unsigned wgs_per_row)
{
    const long long j = 4 * ((long long)(blockIdx.x % wgs_per_row) * 256 + threadIdx.x);
    if (j >= len) return;
    const size_t row = blockIdx.x / wgs_per_row;        // stream * 2 + (q & 1)
    const size_t s = row >> 1;
    const int q = 2 * pair + (int)(row & 1);
    lb_v4f v = {0.f, 0.f, 0.f, 0.f};
    if (keep_mask & (1 << q)) {
        v = *reinterpret_cast<const lb_v4f *>(fresh + row * len + j);
        if (pend && pos + j < len) v += *reinterpret_cast<const lb_v4f *>(pend + (s * 4 + q) * len + pos + j);
    }
    *reinterpret_cast<lb_v4f *>(dst + (s * 4 + q) * len + j) = v;
}

// out[s][ear][f] += gain * (tails[s][ear][pos + f] + tails[s][2 + ear][pos + f]),  f < min(n_frames, len - pos)
__global__ __launch_bounds__(256) void k_conv_lb_tails_add(float *__restrict__ out, long long out_ss, long long out_cs, long long count,
                                                           const float *__restrict__ tails, long long len, long long pos, float gain,
                                                           unsigned wgs_per_row)
{
    const long long f = (long long)(blockIdx.x % wgs_per_row) * 256 + threadIdx.x;
    if (f >= count) return;
    const size_t row = blockIdx.x / wgs_per_row;        // stream * 2 + ear
    const size_t s = row >> 1, ear = row & 1;
    const float t = tails[(s * 4 + ear) * len + pos + f] + tails[(s * 4 + 2 + ear) * len + pos + f];
    float *o = out + s * out_ss + ear * out_cs + f;
    *o = *o + gain * t;
}

hipError_t launch_conv_lb_tails_merge(const float *pend, long long pos, const float *fresh, float *dst, long long len, int pair,
                                      int keep_mask, int n_streams, hipStream_t st)
{
    if (len <= 0 || (len & 511) || (pos & 511) || pos < 0 || n_streams <= 0 || pair < 0 || pair > 1) return hipErrorInvalidValue;
    const unsigned long long wpr = (unsigned long long)((len / 4 + 255) / 256), wgs = wpr * 2ull * (unsigned long long)n_streams;
    if (wgs > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv_lb_tails_merge, dim3((unsigned)wgs), dim3(256), 0, st, pend, pos, fresh, dst, len, pair, keep_mask,
                       (unsigned)wpr);
    return hipGetLastError();
}

hipError_t launch_conv_lb_tails_add(float *out, long long out_ss, long long out_cs, long long n_frames, const float *tails,
                                    long long len, long long pos, float gain, int n_streams, hipStream_t st)
{
    if (len <= 0 || pos < 0 || pos >= len || n_streams <= 0) return hipErrorInvalidValue;
    const long long count = n_frames < len - pos ? n_frames : len - pos;
    if (count <= 0) return hipSuccess;
    const unsigned long long wpr = (unsigned long long)((count + 255) / 256), wgs = wpr * 2ull * (unsigned long long)n_streams;
    if (wgs > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv_lb_tails_add, dim3((unsigned)wgs), dim3(256), 0, st, out, out_ss, out_cs, count, tails, len, pos, gain,
                       (unsigned)wpr);
    return hipGetLastError();
}

hipError_t launch_conv_lb_xhist_append(const float *in, long long in_ss, long long in_cs, float *ring, long long len,
                                       long long head, long long n_frames, int n_streams, hipStream_t st)
{
    if (len <= 0 || (len & 511) || (n_frames & 511) || (head & 511) || head < 0 || head >= 2 * len || n_streams <= 0)
        return hipErrorInvalidValue;
    if ((in_ss | in_cs) & 3 || (reinterpret_cast<uintptr_t>(in) & 15)) return hipErrorInvalidValue;
    const long long count = n_frames < len ? n_frames : len;
    if (count <= 0) return hipSuccess;
    const unsigned long long wpr = (unsigned long long)((count / 4 + 255) / 256), wgs = wpr * 2ull * (unsigned long long)n_streams;
    if (wgs > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv_lb_xhist_append, dim3((unsigned)wgs), dim3(256), 0, st, in, in_ss, in_cs, ring, len, head, n_frames,
                       count, (unsigned)wpr);
    return hipGetLastError();
}

hipError_t launch_conv_lb_xhist_linearise(const float *ring, float *dst, long long len, long long head, int n_streams,
                                          hipStream_t st)
{
    if (len <= 0 || (len & 511) || (head & 511) || head < 0 || head >= 2 * len || n_streams <= 0) return hipErrorInvalidValue;
    const unsigned long long wpr = (unsigned long long)((len / 4 + 255) / 256), wgs = wpr * 2ull * (unsigned long long)n_streams;
    if (wgs > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv_lb_xhist_linearise, dim3((unsigned)wgs), dim3(256), 0, st, ring, dst, len, head, (unsigned)wpr);
    return hipGetLastError();
}

}  // namespace ohs
