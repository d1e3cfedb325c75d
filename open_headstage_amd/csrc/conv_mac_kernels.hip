// conv_mac_kernels.hip -- long impulse responses: the spectral multiply-accumulate of the time-parallel
// path.  Separate translation unit because it wants hipcc's SLP vectoriser (packed f32
// MACs on pairs of accumulators), while the FFT kernels of conv_kernels.hip are built with
// -fno-slp-vectorize: there the vectoriser pairs the real parts of two different complex values and
// pays for it with ~30 % register moves (k_conv_p1: 168 -> 124 VGPRs, 0.97 -> 0.87 ms without it).
#include "kernels.h"
#include "wave_fft.hpp"
#include "tuning.h"

namespace ohs {

typedef float v2f __attribute__((ext_vector_type(2)));

// acc += z * c + conj(m) * d on (re, im) register pairs: four v_pk_fma_f32, every swizzle and sign an
// operand modifier (op_sel picks the half of a source for the low result, op_sel_hi for the high one).
//   re += z.re c.re - z.im c.im + m.re d.re + m.im d.im ;  im += z.re c.im + z.im c.re + m.re d.im - m.im d.re
// hipcc's own code for the same expression is 2.6 VALU instructions per FMA pair (separate multiplies
// and adds plus register moves for the SLP pairing); this is 1.
__device__ __forceinline__ void cmac_z_c_conjm_d(v2f &acc, v2f z, v2f c, v2f m, v2f d)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(z), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(acc) : "v"(z), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(m), "v"(d));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "+v"(acc) : "v"(m), "v"(d));
}

// position in the bin layout holding bin N-k of the bin held at position b
__device__ __forceinline__ int mirror_position(int b)
{
    const int r = b >> 6, L = b & 63;
    const int j = r & 3, q = r >> 2;
    if (j) return (4 * (3 - q) + (4 - j)) * 64 + (63 - L);
    if (L) return (4 * (3 - q)) * 64 + (64 - L);
    return (4 * ((4 - q) & 3)) * 64;
}

// M: thread = one layout position; workgroup = (quarter of the spectrum, TB-block time tile, stream).
// Register tile: TB output blocks x PI partitions per window = TB*PI complex MAC pairs from
// 2(TB+PI-1) history loads + 2 PI spectrum loads.  The kernel is bound by L2 -> L1 traffic (every load
// is an L2 hit once the XCD mapping below is in place, L1 is far too small), so bigger tiles win until
// registers run out -- config 4, whole convolution step:
//   compiler-scheduled MAC, (TB,PI) = (8,8): 0.372 ms | explicit v_pk_fma_f32 (8,8): 0.334 | (8,16): 0.313
//   (16,8): 0.286 | (16,16): 0.277 (137 VGPRs) | (32,8): 0.265 | (32,16): 0.256 (254 VGPRs)
// Both (16,16) and (32,16) are built; the launcher takes TB = 32 where the segment's block count fills its tiles well
// enough (launch_conv_tp_mac): short segments (many streams) waste less of a tile with TB = 16.
// (A variant with one thread per position PAIR {b, mirror(b)} halves the history reads; with the
// compiler-scheduled MAC it needed 256 VGPRs and ran at 0.69 ms.)
#ifndef OHS_MAC_TB
#define OHS_MAC_TB 0                // 0: chosen per launch between 16 and 32; 16 / 32: that tile only (experiments)
#endif
#ifndef OHS_MAC_PI
#define OHS_MAC_PI 16
#endif
constexpr int kPI = OHS_MAC_PI;     // partitions per register window (Ppad is a multiple of 16, so 8 or 16)
__device__ __forceinline__ void conv_tp_w1_job(const ConvTpArgs &A, int wg);

template <int kTB>                  // output blocks per register tile
__global__ __launch_bounds__(256) void k_conv_tp_mac(const ConvTpArgs A)
{
    ohs_set_fp_mode(A.fp_mode);
    // the trailing workgroups of the grid compute W1 (below): one launch instead of two, and they fill the slots the
    // MAC's last round leaves empty
    if ((int)blockIdx.x >= A.main_wgs) { conv_tp_w1_job(A, (int)blockIdx.x - A.main_wgs); return; }
    // XCD-aware mapping.  Workgroups go to the 8 XCDs round-robin by linear id and every XCD has its own
    // 4 MiB L2; all workgroups of one stream re-read that stream's history ring (P + n_mac spectra,
    // 2.3 MiB on config 4), so a stream's workgroups are given ids that are congruent mod 8: its ring
    // is then fetched into ONE L2 instead of eight.
    const unsigned tiles = (unsigned)(A.mac_tiles < 0 ? -A.mac_tiles : A.mac_tiles);
    const unsigned wps = (kFft / 256) * tiles;                  // workgroups per stream
    size_t s;
    unsigned widx;
    if (A.mac_tiles > 0) {
        const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        s = (size_t)(slot / wps) * 8 + xcd;
        widx = slot % wps;
    } else {                    // plain stream-major order (experiments: mac_no_xcd, for comparison)
        s = blockIdx.x / wps;
        widx = blockIdx.x % wps;
    }
    if (s >= (size_t)A.n_streams) return;
    const int b = (int)(widx & (kFft / 256 - 1)) * 256 + threadIdx.x;
    const int t0 = (int)(widx / (kFft / 256)) * kTB;
    const int mb = mirror_position(b);
    const float2 *hist = A.hist + s * (size_t)A.cap * kFft;
    const unsigned long long mask = (unsigned long long)(A.cap - 1);
    const unsigned long long base = A.cnt + (unsigned long long)t0;

    v2f acc[kTB];
#pragma unroll
    for (int t = 0; t < kTB; ++t) acc[t] = (v2f){0.f, 0.f};

    for (int i0 = 0; i0 < A.Ppad; i0 += kPI) {
        // window of blocks u = t - i, t in [0,TB), i in [i0, i0+PI): w = (t - i) + i0 + PI - 1 in [0, TB+PI-1)
        v2f zw[kTB + kPI - 1], zm[kTB + kPI - 1];
#pragma unroll
        for (int w = 0; w < kTB + kPI - 1; ++w) {
            const unsigned long long u = base + (unsigned long long)(long long)(w - (kPI - 1) - i0);
            const v2f *hs = reinterpret_cast<const v2f *>(hist + (size_t)(u & mask) * kFft);
            zw[w] = hs[b];
            zm[w] = hs[mb];
        }
#pragma unroll
        for (int ii = 0; ii < kPI; ++ii) {
            const v2f c = reinterpret_cast<const v2f *>(A.CD)[((size_t)(i0 + ii) * 2 + 0) * kFft + b];
            const v2f d = reinterpret_cast<const v2f *>(A.CD)[((size_t)(i0 + ii) * 2 + 1) * kFft + b];
#pragma unroll
            for (int t = 0; t < kTB; ++t) cmac_z_c_conjm_d(acc[t], zw[t - ii + kPI - 1], c, zm[t - ii + kPI - 1], d);
        }
    }
#pragma unroll
    for (int t = 0; t < kTB; ++t)
        if (t0 + t < A.n_mac)
            reinterpret_cast<v2f *>(A.W)[(s * (size_t)A.n_mac + (size_t)(t0 + t)) * kFft + b] = acc[t];
}

// The segment's LAST block additionally needs its L-speaker part W1 = sum_i X_l A_i, with
// X_l = (Z + conj(Zm))/2 and A_i = H_lsl + j H_lsr = C_i + D_i, so that the four per-path overlaps of
// the state at rest can be rebuilt (W2 = W - W1).  One block per stream: workgroups of their own (the trailing
// ones of k_conv_tp_mac's grid, three per stream), so the main MAC loop carries no per-iteration test for it (that
// test cost the main kernel 55 %); until round 3 they were a kernel of their own.
__device__ __forceinline__ void conv_tp_w1_job(const ConvTpArgs &A, int wg)
{
    const int per_stream = (A.n_pairs + 255) / 256;
    const int pi = (wg % per_stream) * 256 + (int)threadIdx.x;
    if (pi >= A.n_pairs) return;
    const int b = A.pairs[pi];
    const int mb = mirror_position(b);
    const size_t s = (size_t)(wg / per_stream);
    const float2 *hist = A.hist + s * (size_t)A.cap * kFft;
    const unsigned long long mask = (unsigned long long)(A.cap - 1);
    const unsigned long long last = A.cnt + (unsigned long long)(A.n_mac - 1);
    float2 acc1b = make_float2(0.f, 0.f), acc1m = make_float2(0.f, 0.f);
    for (int i0 = 0; i0 < A.Ppad; i0 += 8) {       // Ppad is a multiple of 8: 48 loads in flight per step
        float2 zz[8], mm[8], cbv[8], dbv[8], cmv[8], dmv[8];
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const int i = i0 + ii;
            const float2 *hs = hist + (size_t)((last - (unsigned long long)i) & mask) * kFft;
            zz[ii] = hs[b]; mm[ii] = hs[mb];
            cbv[ii] = A.CD[((size_t)i * 2 + 0) * kFft + b]; dbv[ii] = A.CD[((size_t)i * 2 + 1) * kFft + b];
            cmv[ii] = A.CD[((size_t)i * 2 + 0) * kFft + mb]; dmv[ii] = A.CD[((size_t)i * 2 + 1) * kFft + mb];
        }
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
        const float2 z = zz[ii], m = mm[ii], cb = cbv[ii], db = dbv[ii], cm = cmv[ii], dm = dmv[ii];
        const float2 xlb = make_float2(0.5f * (z.x + m.x), 0.5f * (z.y - m.y));
        const float2 ab = make_float2(cb.x + db.x, cb.y + db.y);
        acc1b.x += xlb.x * ab.x - xlb.y * ab.y;
        acc1b.y += xlb.x * ab.y + xlb.y * ab.x;
        const float2 xlm = make_float2(0.5f * (m.x + z.x), 0.5f * (m.y - z.y));
        const float2 am = make_float2(cm.x + dm.x, cm.y + dm.y);
        acc1m.x += xlm.x * am.x - xlm.y * am.y;
        acc1m.y += xlm.x * am.y + xlm.y * am.x;
        }
    }
    A.W1[s * (size_t)kFft + b] = acc1b;
    if (b != mb) A.W1[s * (size_t)kFft + mb] = acc1m;
}

// time tile of a segment of n blocks: 32 x 16 runs a block 11 % faster than 16 x 16 (config 4: 0.269 -> 0.250 ms per step
// in all) but rounds the segment up to a multiple of 32.  Short calls (a many-stream host that calls with the plugin's
// 1024-frame block: 2 blocks) take tiles of 4 or 8 blocks: a 16-block tile would compute 16 outputs to store 2.
static int mac_time_tile(int n)
{
    if (OHS_MAC_TB != 0) return OHS_MAC_TB;
    if (n <= 4) return 4;
    if (n <= 8) return 8;
    const long long c32 = (long long)((n + 31) / 32) * 32 * 89, c16 = (long long)((n + 15) / 16) * 16 * 100;
    return c32 < c16 ? 32 : 16;
}

hipError_t launch_conv_tp_mac(const ConvTpArgs &a, hipStream_t st)
{
    if (a.n_mac <= 0) return hipSuccess;
    ConvTpArgs m = a;
    const int kTB = mac_time_tile(a.n_mac);
    m.mac_tiles = (a.n_mac + kTB - 1) / kTB;
    const unsigned wps = (unsigned)(kFft / 256) * (unsigned)m.mac_tiles;
    const unsigned groups = ((unsigned)a.n_streams + 7u) / 8u;
    const bool xcd_map = !tuning().mac_no_xcd;
    if (!xcd_map) m.mac_tiles = -m.mac_tiles;
    m.main_wgs = (int)(groups * 8u * wps);
    const unsigned w1_wgs = (unsigned)((a.n_pairs + 255) / 256) * (unsigned)a.n_streams;
    const dim3 grid((unsigned)m.main_wgs + w1_wgs);
    switch (kTB) {
    case 4: hipLaunchKernelGGL(k_conv_tp_mac<4>, grid, dim3(256), 0, st, m); break;
    case 8: hipLaunchKernelGGL(k_conv_tp_mac<8>, grid, dim3(256), 0, st, m); break;
    case 16: hipLaunchKernelGGL(k_conv_tp_mac<16>, grid, dim3(256), 0, st, m); break;
    case 32: hipLaunchKernelGGL(k_conv_tp_mac<32>, grid, dim3(256), 0, st, m); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// O: gated contributions of the blocks the ring held before the call (kernels.h: ConvTpOldArgs).  Old block -d may be
// seen by path p iff d <= since[p] (the reference resets only THAT path's history, convolution.rs:135-138); it meets
// partition i = t + d of output block t.  The combined spectra are built per (d, i) from the visible paths exactly as
// k_build_cd_multi builds them from all four: C = (A - jB)/2, D = (A + jB)/2, A = H0 + jH1, B = H2 + jH3.
__global__ __launch_bounds__(256) void k_conv_tp_old(const ConvTpOldArgs A)
{
    ohs_set_fp_mode(A.fp_mode);
    const int b = (int)(blockIdx.x & (kFft / 256 - 1)) * 256 + threadIdx.x;
    const int t = (int)(blockIdx.x / (kFft / 256)) % A.n_old;
    const size_t s = (size_t)(blockIdx.x / (kFft / 256)) / (size_t)A.n_old;
    if (s >= (size_t)A.n_streams) return;
    const int mb = mirror_position(b);
    const float2 *hist = A.hist + s * (size_t)A.cap * kFft;
    const unsigned long long mask = (unsigned long long)(A.cap - 1);
    float2 acc = make_float2(0.f, 0.f);
    bool any = false;
    const int d_end = A.Pmax - t;               // i = t + d <= Pmax - 1
    for (int d = 1; d < d_end; ++d) {
        const int i = t + d;
        const bool g0 = i < A.P[0] && (long long)d <= A.since[0];
        const bool g1 = i < A.P[1] && (long long)d <= A.since[1];
        const bool g2 = i < A.P[2] && (long long)d <= A.since[2];
        const bool g3 = i < A.P[3] && (long long)d <= A.since[3];
        if (!(g0 || g1 || g2 || g3)) continue;      // (wave-uniform)
        any = true;
        const float2 *hs = hist + (size_t)((A.cnt_old - (unsigned long long)d) & mask) * kFft;
        const float2 z = hs[b], m = hs[mb];
        // branch-free over the paths: the partition index is clamped into the path's own range, a gated-off path's
        // spectrum is multiplied by 0 (block_mac in conv_kernels.hip does the same)
        const float f0 = g0 ? 1.0f : 0.0f, f1 = g1 ? 1.0f : 0.0f, f2 = g2 ? 1.0f : 0.0f, f3 = g3 ? 1.0f : 0.0f;
        const float2 h0 = A.H[0][(size_t)(i < A.P[0] ? i : A.P[0] - 1) * kFft + b];
        const float2 h1 = A.H[1][(size_t)(i < A.P[1] ? i : A.P[1] - 1) * kFft + b];
        const float2 h2 = A.H[2][(size_t)(i < A.P[2] ? i : A.P[2] - 1) * kFft + b];
        const float2 h3 = A.H[3][(size_t)(i < A.P[3] ? i : A.P[3] - 1) * kFft + b];
        const float2 A_ = make_float2(h0.x * f0 - h1.y * f1, h0.y * f0 + h1.x * f1);
        const float2 B_ = make_float2(h2.x * f2 - h3.y * f3, h2.y * f2 + h3.x * f3);
        const float2 c = make_float2(0.5f * (A_.x + B_.y), 0.5f * (A_.y - B_.x));
        const float2 dd = make_float2(0.5f * (A_.x - B_.y), 0.5f * (A_.y + B_.x));
        acc.x += z.x * c.x - z.y * c.y + m.x * dd.x + m.y * dd.y;
        acc.y += z.x * c.y + z.y * c.x + m.x * dd.y - m.y * dd.x;
    }
    if (any) {
        float2 *w = A.W + (s * (size_t)A.n_mac + (size_t)t) * kFft + b;
        const float2 o = *w;
        *w = make_float2(o.x + acc.x, o.y + acc.y);
    }
}

hipError_t launch_conv_tp_old(const ConvTpOldArgs &a, hipStream_t st)
{
    if (a.n_old <= 0 || a.n_streams <= 0) return hipSuccess;
    if (a.n_old > a.n_mac) return hipErrorInvalidValue;
    const unsigned long long wgs = (unsigned long long)(kFft / 256) * (unsigned long long)a.n_old * (unsigned long long)a.n_streams;
    if (wgs > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv_tp_old, dim3((unsigned)wgs), dim3(256), 0, st, a);
    return hipGetLastError();
}

// host helper: the leaders of the 514 position pairs {b, mirror(b)}, ordered so that consecutive
// threads touch consecutive addresses as far as possible (ascending b)
int conv_tp_build_pairs(int *out)
{
    int n = 0;
    for (int b = 0; b < kFft; ++b) {
        const int r = b >> 6, L = b & 63;
        const int j = r & 3, q = r >> 2;
        int mb;
        if (j) mb = (4 * (3 - q) + (4 - j)) * 64 + (63 - L);
        else if (L) mb = (4 * (3 - q)) * 64 + (64 - L);
        else mb = (4 * ((4 - q) & 3)) * 64;
        if (b <= mb) out[n++] = b;
    }
    return n;
}

}  // namespace ohs
