// conv_os_kernels.hip -- every path one partition (taps <= 512), batch mode: overlap-SAVE with a 2048-point transform and a
// hop of 1536 frames, built from the wave's 1024-point transform by one radix-2 step that needs no exchange.
//
// Why: k_conv_p1 (block 512 / FFT 1024, overlap-add, conv_kernels.hip) runs at the board's power cap -- its time follows
// the energy per output sample: LDS exchange bytes, table reads, vector instructions.  A 2048-point
// window that yields 1536 output samples costs two 1024-point transforms each way instead of three, and two thirds of the
// spectral products: a third less of everything that is not HBM traffic, for one more twiddle stage.
//
// How, without a new transform.  Decimation in time: with e[m] = z[2m], o[m] = z[2m + 1] (z = x_l + j x_r, the packed
// stereo window), E = FFT1024(e), O = FFT1024(o) and w = exp(-2 pi i / 2048):
//     Z[k] = E[k] + w^k O[k],   Z[k + 1024] = E[k] - w^k O[k]                                   k = 0 .. 1023
// Both 1024-point spectra sit in the SAME lanes and registers (the paired bin layout of wave_fft.hpp), so the step is
// lane-local.  The mirror bin of the packed two-channel product, Z[2048 - k], is bin 1024 + (1024 - k) = the "hi" value at
// the 1024-mirror of k -- and the paired layout keeps bin 1024 - k in the lane that holds k.  Hence
//     W[k]        = Z[k] C[k] + conj(Z[2048 - k]) D[k]:   W_lo[k] = Z_lo[k] C_lo[k] + conj(Z_hi[m]) D_lo[k]
//                                                         W_hi[k] = Z_hi[k] C_hi[k] + conj(Z_lo[m]) D_hi[k],  m = (1024 - k) % 1024
// (k = 0: bins 0 and 1024 mirror themselves: lo with lo, hi with hi), and decimation in frequency on the way back:
//     y[2m] = IFFT1024(W_lo + W_hi)[m],   y[2m + 1] = IFFT1024((W_lo - W_hi) conj(w^k))[m].
// C = (A - jB)/2, D = (A + jB)/2, A = H_lsl + j H_lsr, B = H_rsl + j H_rsr as in k_conv_p1, from the 2048-point spectra of
// the zero-padded impulse responses, 1/2048 folded in.  Overlap-save: window h covers frames [1536 h - 512, 1536 h + 1536),
// its outputs [512, 2048) are frames [1536 h, 1536 h + 1536) of convolve_path_partitioned's result
// (src/dsp/convolution.rs:236-289) summed per ear (:228-231) -- equal to the reference up to f32 rounding, NOT bit for
// bit (the block-512 kernel keeps the golden hash; ohs_batch_set_conv_plan selects).
//
// Lane L holds window samples 128 a + 2 L and 128 a + 2 L + 1 of both channels in register a: every global access is an
// 8-byte, lane-contiguous load or store; the window's first 512 samples are the previous window's last (registers, no
// re-read), so the kernel moves 16 B per frame like k_conv_p1.
#include "kernels.h"
#include "wave_fft.hpp"

#include <atomic>

namespace ohs {

// Launch shape and register plan (constants of the product; experiments builds may override them):
//   OHS_OS_WAVES     waves per workgroup = per CU.  12 = three per SIMD: 168 registers per lane, and the LDS plan below still
//                    fits (48.5 KiB of tables + 12 slabs = 152.5 KiB).  With 8 (two per SIMD, 256 registers) the kernel is
//                    issue-latency-bound: at the board's power cap it clocks 2.07 GHz against the block-512 kernel's 1.82
//                    and needs 11 % more cycles for a quarter less work -- a tie (profiles/r04_power_conv_plans.txt).
//   OHS_OS_PARK      1: the spectral product's results for slots {0, 1} (16 complex values per lane) wait in the wave's
//                    exchange slab -- idle between the transforms -- while slots {2, 3} go through.  Without it hipcc
//                    needs ~290 registers for the product phase and the 12-wave kernel spills 130 scratch instructions
//                    per hop (0.92 ms per headline-sized step); with it the steady hop has no spill (0.486 ms).
//   OHS_OS_PREFETCH  where the next hop's 24 loads are issued: 0 = behind the spectral product, 1 = between the two inverse
//                    transforms, 2 = behind both (in front of the stores).  0 keeps 48 more registers alive through both
//                    inverse transforms; 2 is the one that fits 168 registers without a spill in the steady hop.
//                    3 = the first OHS_OS_SPLIT register pairs between the inverse transforms, the rest behind both
//                    (4 scratch instructions per hop): 3 % faster than 2 at 256 streams (one round of waves, the
//                    loads' latency shows), equal or 1 % slower at 2 048 -- where the library picks this plan; 1 alike.
#if !defined(OHS_OS_PREFETCH) || !defined(OHS_EXPERIMENTS)
#undef OHS_OS_PREFETCH
#define OHS_OS_PREFETCH 2
#endif
#if !defined(OHS_OS_SPLIT) || !defined(OHS_EXPERIMENTS)
#undef OHS_OS_SPLIT
#define OHS_OS_SPLIT 6
#endif
#if !defined(OHS_OS_PARK) || !defined(OHS_EXPERIMENTS)
#undef OHS_OS_PARK
#define OHS_OS_PARK 1
#endif
#if !defined(OHS_OS_WAVES) || !defined(OHS_EXPERIMENTS)
#undef OHS_OS_WAVES
#define OHS_OS_WAVES 12
#endif
constexpr int kOsWaves = OHS_OS_WAVES;
constexpr int kOsHop = 1536;
// LDS: twiddle tables + (C, D) lo / hi [2][16][64] float4 + w2048 [16][64] float2 + one FFT slab per wave
constexpr size_t kOsCdComplex = 2 * 2 * kFft;      // float4 = 2 float2
constexpr size_t kOsLdsBytes = (kTabComplex + kOsCdComplex + kFft + kOsWaves * (size_t)kWaveLdsComplex) * sizeof(float2);
static_assert(kOsLdsBytes <= 160 * 1024, "k_conv_p1_os: LDS plan");

// 1024-point bin held by paired register i of `lane`
__host__ __device__ __forceinline__ constexpr int os_bin(int lane, int i)
{
    const int q = paired_to_natural(lane, i);
    const int rn = q >> 6, ln = q & 63;
    return 4 * ln + (rn & 3) + 256 * (rn >> 2);
}
static_assert(os_bin(32, 12) == 0 && os_bin(32, 14) == 512, "paired layout: bins 0 and 512 live in lane 32");

static hipError_t os_allow_large_lds(const void *fn, size_t bytes, std::atomic<unsigned long long> &done_mask)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done_mask.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done_mask.fetch_or(bit, std::memory_order_release);
    return e;
}

// the audio is read once and written once: non-temporal 8-byte accesses (the builtins want a native vector type)
typedef float os_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 os_ld_nt(const float *p)
{
    const os_v2f v = __builtin_nontemporal_load(reinterpret_cast<const os_v2f *>(p));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ void os_st_nt(float *p, float a, float b)
{
    const os_v2f v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<os_v2f *>(p));
}

// full (not zero-padded) window in, spectrum in the paired layout out
__device__ __forceinline__ void os_fft_fwd(float2 (&v)[16], float2 *lds, const float2 *tab, int lane, const PairedPlan &pl)
{
    wave_fft_fwd_head<false>(v, lds, tab, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * g + c] = lds_ld(&lds[pl.t2[g] + 65 * c]);
    }
    wave_lds_sync();
#pragma unroll
    for (int g = 0; g < 4; ++g) dft4<false>(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}

// host: w2048^bin for every paired position, f64 -> f32
void os_build_twiddles(float2 *out)
{
    for (int i = 0; i < 16; ++i)
        for (int lane = 0; lane < 64; ++lane) {
            const double ang = -2.0 * 3.14159265358979323846264338327950288 * (double)os_bin(lane, i) / 2048.0;
            out[i * 64 + lane] = make_float2((float)cos(ang), (float)sin(ang));
        }
}

// ---------------------------------------------------------------------------------
// tables: the four impulse responses (time domain, 512 floats each, zero-padded) -> (C, D) lo / hi
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_os_tables(const float *__restrict__ irt, const float2 *__restrict__ tw,
                                                   const float2 *__restrict__ tw2048, float4 *__restrict__ cd_out)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    float2 *hs = smem + kTabComplex;                    // [4 paths][2 lo/hi][1024]
    float2 *slabs = hs + 4 * 2 * kFft;
    fill_twiddle_tables(tab, tw, threadIdx.x, 256);
    __syncthreads();
    const int p = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float2 *lds = slabs + p * kWaveLdsComplex;
    const PairedPlan plan = paired_plan(lane);
    const float *h = irt + p * kBlock;
    float2 E[16], O[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const int n = 128 * a + 2 * lane;               // taps 2m, 2m + 1 of the 2048-point window
        E[a] = make_float2(n < kBlock ? h[n] : 0.0f, 0.0f);
        O[a] = make_float2(n + 1 < kBlock ? h[n + 1] : 0.0f, 0.0f);
    }
    os_fft_fwd(E, lds, tab, lane, plan);
    os_fft_fwd(O, lds, tab, lane, plan);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float2 t = cmul(O[i], tw2048[i * 64 + lane]);
        hs[(p * 2 + 0) * kFft + i * 64 + lane] = cadd(E[i], t);
        hs[(p * 2 + 1) * kFft + i * 64 + lane] = csub(E[i], t);
    }
    __syncthreads();
    const float sc = 0.5f / 2048.0f;                    // the 1/2 of C, D and the 1/N of the inverse transform (powers of two)
    for (int q = threadIdx.x; q < 2 * kFft; q += 256) {
        const int half = q >> 10, pos = q & (kFft - 1);
        const float2 a0 = hs[(0 * 2 + half) * kFft + pos], a1 = hs[(1 * 2 + half) * kFft + pos];
        const float2 a2 = hs[(2 * 2 + half) * kFft + pos], a3 = hs[(3 * 2 + half) * kFft + pos];
        const float2 A_ = make_float2(a0.x - a1.y, a0.y + a1.x);
        const float2 B_ = make_float2(a2.x - a3.y, a2.y + a3.x);
        cd_out[q] = make_float4(sc * (A_.x + B_.y), sc * (A_.y - B_.x), sc * (A_.x - B_.y), sc * (A_.y + B_.x));
    }
}

hipError_t launch_os_tables(const float *d_irt, const float2 *tw, const float2 *tw2048, float4 *cd_out, hipStream_t st)
{
    const size_t shmem = (kTabComplex + 4 * 2 * kFft + 4 * (size_t)kWaveLdsComplex) * sizeof(float2);
    static std::atomic<unsigned long long> lds_ok{0};
    const hipError_t e = os_allow_large_lds(reinterpret_cast<const void *>(k_os_tables), shmem, lds_ok);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_os_tables, dim3(1), dim3(256), shmem, st, d_irt, tw, tw2048, cd_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// the hop loop
// ---------------------------------------------------------------------------------
// hops of chunk ck: the H hops of the launch cut into `chunks` contiguous ranges
__host__ __device__ __forceinline__ int os_chunk_begin(int ck, int chunks, int H) { return (int)((long long)ck * H / chunks); }

struct OsWave {             // wave-uniform context of the hop loop
    const float *in_l, *in_r;
    float *out_l, *out_r;
    const float4 *cd;       // LDS
    const float2 *tw2;      // LDS
    float2 *lds;
    const float2 *tab;
    unsigned n_frames, v_last;
    int H, h1, s;
    bool last_chunk;
};

// new frames of hop h -> nl / nr; GUARD: frames beyond the call are zeros and are not read
template <bool GUARD, int A0 = 0, int A1 = 12>
__device__ __forceinline__ void os_load_new(const OsWave &w, int h, int lane, float2 (&nl)[12], float2 (&nr)[12])
{
    const unsigned base = (unsigned)h * kOsHop + 2u * (unsigned)lane;
#pragma unroll
    for (int a = A0; a < A1; ++a) {
        const bool ok = !GUARD || (unsigned long long)h * kOsHop + 128ull * a < w.n_frames;       // wave-uniform
        nl[a] = ok ? os_ld_nt(w.in_l + base + 128 * a) : make_float2(0.f, 0.f);
        nr[a] = ok ? os_ld_nt(w.in_r + base + 128 * a) : make_float2(0.f, 0.f);
    }
}

// One hop.  GENERIC = false: the steady state -- this hop and the next are whole and inside the wave's range, nothing but
// the transforms, the product, 24 loads and 24 stores.  GENERIC = true: the call's first hop (state in), a chunk's last
// hops (ragged end, state out).
template <bool GENERIC>
__device__ __forceinline__ void os_hop(const ConvOsArgs &A, const OsWave &w, int h, int lane, const PairedPlan &plan,
                                       float2 (&pl)[4], float2 (&pr)[4], float2 (&nl)[12], float2 (&nr)[12])
{
    float2 E[16], O[16];
#pragma unroll
    for (int a = 0; a < 4; ++a) { E[a] = make_float2(pl[a].x, pr[a].x); O[a] = make_float2(pl[a].y, pr[a].y); }
#pragma unroll
    for (int a = 0; a < 12; ++a) { E[a + 4] = make_float2(nl[a].x, nr[a].x); O[a + 4] = make_float2(nl[a].y, nr[a].y); }
    const bool final_hop = GENERIC && w.last_chunk && h == w.h1 - 1;
    // v' = this hop's valid new frames in blocks of 512 (0: the window of zeros behind the call)
    const unsigned vq = (h == w.H) ? 0u : w.v_last / kBlock;
    if (GENERIC && final_hop) {
        // the call's last 512 input frames (what conv_materialise_state rebuilds the per-path overlaps from): window
        // samples [v', v' + 512)
        float2 *li_l = reinterpret_cast<float2 *>(A.last_in + (size_t)w.s * (2 * kBlock)) + lane;
        float2 *li_r = li_l + kBlock / 2;
        if (vq == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a) { li_l[64 * a] = pl[a]; li_r[64 * a] = pr[a]; }
        } else if (vq == 1) {
#pragma unroll
            for (int a = 0; a < 4; ++a) { li_l[64 * a] = nl[a]; li_r[64 * a] = nr[a]; }
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) { li_l[64 * a] = nl[a + 4]; li_r[64 * a] = nr[a + 4]; }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) { pl[a] = nl[a + 8]; pr[a] = nr[a + 8]; }       // the next window's first 512 samples

    // (one transform after the other: the two are independent and hipcc would interleave them -- twice the temporaries)
    os_fft_fwd(E, w.lds, w.tab, lane, plan);
    __builtin_amdgcn_sched_barrier(0);
    os_fft_fwd(O, w.lds, w.tab, lane, plan);
    __builtin_amdgcn_sched_barrier(0);

    // radix-2 step, spectral product, radix-2 step back -- per paired register, mirrors from the lane's own registers
    // (wave_fft.hpp: paired_mirror_reg).  The 2048-point spectrum is never alive as a whole: slots {0, 1} mirror each
    // other register by register (r <-> 7 - r), so they go through as four pairs; slots {2, 3} mirror among themselves
    // (lane 32 holds the self-mirrored groups there) and go through as one set of eight.
    const bool lane32 = lane == 32;
    auto product = [&](int i, float2 xl, float2 xh, float2 ml, float2 mh, float2 tw) {
        float2 c, d;
        lds_ld2(w.cd + i * 64 + lane, c, d);
        float2 wl, wh;
        wl.x = fmaf(ml.y, d.y, fmaf(ml.x, d.x, fmaf(-xl.y, c.y, xl.x * c.x)));
        wl.y = fmaf(-ml.y, d.x, fmaf(ml.x, d.y, fmaf(xl.y, c.x, xl.x * c.y)));
        lds_ld2(w.cd + kFft + i * 64 + lane, c, d);
        wh.x = fmaf(mh.y, d.y, fmaf(mh.x, d.x, fmaf(-xh.y, c.y, xh.x * c.x)));
        wh.y = fmaf(-mh.y, d.x, fmaf(mh.x, d.y, fmaf(xh.y, c.x, xh.x * c.y)));
        E[i] = cadd(wl, wh);
        O[i] = cmulc(csub(wl, wh), tw);
    };
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = r, m = 7 - r;
        const float2 twi = lds_ld(&w.tw2[i * 64 + lane]), twm = lds_ld(&w.tw2[m * 64 + lane]);
        const float2 ti = cmul(O[i], twi), tm = cmul(O[m], twm);
        const float2 xli = cadd(E[i], ti), xhi = csub(E[i], ti), xlm = cadd(E[m], tm), xhm = csub(E[m], tm);
        // Z[2048 - k] = hi at the 1024-mirror of k;  Z[2048 - (k + 1024)] = lo at the 1024-mirror of k
        product(i, xli, xhi, xhm, xlm, twi);
        product(m, xlm, xhm, xhi, xli, twm);
    }
    // the results of slots {0, 1} wait in the wave's exchange slab (idle between the transforms) while slots {2, 3} go
    // through: 32 registers less alive there -- what three waves per SIMD (168 registers) need
#if OHS_OS_PARK
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) { lds_st(&w.lds[j * 64 + lane], E[j]); lds_st(&w.lds[(8 + j) * 64 + lane], O[j]); }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
    {
        float2 Xl[8], Xh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 t = cmul(O[8 + j], lds_ld(&w.tw2[(8 + j) * 64 + lane]));
            Xl[j] = cadd(E[8 + j], t);
            Xh[j] = csub(E[8 + j], t);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = 8 + j, g = i >> 2, r = i & 3;
            const int mo = paired_mirror_reg(g, r, false) - 8, ms = paired_mirror_reg(g, r, true) - 8;
            // lane 32's slots 2, 3 hold self-mirrored groups (bins 128 + 256 r; 0, 256, 512, 768)
            float2 ml = make_float2(lane32 ? Xh[ms].x : Xh[mo].x, lane32 ? Xh[ms].y : Xh[mo].y);
            float2 mh = make_float2(lane32 ? Xl[ms].x : Xl[mo].x, lane32 ? Xl[ms].y : Xl[mo].y);
            if (i == 12) {      // bin 0 (lane 32): bins 0 and 1024 mirror themselves
                ml = make_float2(lane32 ? Xl[j].x : ml.x, lane32 ? Xl[j].y : ml.y);
                mh = make_float2(lane32 ? Xh[j].x : mh.x, lane32 ? Xh[j].y : mh.y);
            }
            product(i, Xl[j], Xh[j], ml, mh, lds_ld(&w.tw2[i * 64 + lane]));
        }
    }
#if OHS_OS_PARK
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) { E[j] = lds_ld(&w.lds[j * 64 + lane]); O[j] = lds_ld(&w.lds[(8 + j) * 64 + lane]); }
    wave_lds_sync();
#endif
#if OHS_OS_PREFETCH == 0
    if (!GENERIC) os_load_new<false>(w, h + 1, lane, nl, nr);
    else if (h + 1 < w.h1) os_load_new<true>(w, h + 1, lane, nl, nr);
#endif
    __builtin_amdgcn_sched_barrier(0);
    wave_fft_inv_paired(E, w.lds, w.tab, lane, plan);       // y[2m]:     re = left ear, im = right ear
    __builtin_amdgcn_sched_barrier(0);
#if OHS_OS_PREFETCH == 1
    if (!GENERIC) os_load_new<false>(w, h + 1, lane, nl, nr);
    else if (h + 1 < w.h1) os_load_new<true>(w, h + 1, lane, nl, nr);
    __builtin_amdgcn_sched_barrier(0);
#elif OHS_OS_PREFETCH == 3      // the first OHS_OS_SPLIT register pairs here, the rest behind the second inverse transform
    if (!GENERIC) os_load_new<false, 0, OHS_OS_SPLIT>(w, h + 1, lane, nl, nr);
    else if (h + 1 < w.h1) os_load_new<true, 0, OHS_OS_SPLIT>(w, h + 1, lane, nl, nr);
    __builtin_amdgcn_sched_barrier(0);
#endif
    wave_fft_inv_paired(O, w.lds, w.tab, lane, plan);       // y[2m + 1]
    __builtin_amdgcn_sched_barrier(0);
    // the next hop's new frames are requested before this hop's stores are issued (vector-memory operations retire in
    // order: loads issued behind the stores could not be consumed before every store had been acknowledged)
#if OHS_OS_PREFETCH == 2
    if (!GENERIC) os_load_new<false>(w, h + 1, lane, nl, nr);
    else if (h + 1 < w.h1) os_load_new<true>(w, h + 1, lane, nl, nr);
    __builtin_amdgcn_sched_barrier(0);
#elif OHS_OS_PREFETCH == 3
    if (!GENERIC) os_load_new<false, OHS_OS_SPLIT, 12>(w, h + 1, lane, nl, nr);
    else if (h + 1 < w.h1) os_load_new<true, OHS_OS_SPLIT, 12>(w, h + 1, lane, nl, nr);
    __builtin_amdgcn_sched_barrier(0);
#endif

    if (GENERIC && h == 0) {    // the state the call starts from: the overlap of the frames before it, added to frames [0, 512)
        if (A.merged_in) {
            const float4 *t4 = reinterpret_cast<const float4 *>(A.merged_in + (size_t)w.s * (8 * 64)) + lane;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float4 t = t4[64 * a];
                E[a + 4].x += t.x; E[a + 4].y += t.y; O[a + 4].x += t.z; O[a + 4].y += t.w;
            }
        } else {
            const float4 *t4 = reinterpret_cast<const float4 *>(A.tails + (size_t)w.s * (2 * 8 * 64)) + lane;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float4 t1 = t4[64 * a], t2 = t4[256 + 64 * a];        // (lsl + j lsr) and (rsl + j rsr) tails
                E[a + 4].x += t1.x + t2.x; E[a + 4].y += t1.y + t2.y; O[a + 4].x += t1.z + t2.z; O[a + 4].y += t1.w + t2.w;
            }
        }
    }
    const unsigned obase = (unsigned)h * kOsHop + 2u * (unsigned)lane;
#pragma unroll
    for (int a = 0; a < 12; ++a) {
        if (!GENERIC || (unsigned long long)h * kOsHop + 128ull * a < w.n_frames) {       // wave-uniform
            os_st_nt(w.out_l + obase + 128 * a, E[a + 4].x * A.gain, O[a + 4].x * A.gain);
            os_st_nt(w.out_r + obase + 128 * a, E[a + 4].y * A.gain, O[a + 4].y * A.gain);
        }
    }
    if (GENERIC && final_hop) {
        // the merged overlap behind the call's last frame: outputs [512 + v', 1024 + v') of this window, in the
        // [8][64] (sample 64 a + lane) form k_conv_p1 reads and writes
        float4 *mo = reinterpret_cast<float4 *>(A.merged_out + (size_t)w.s * (8 * 64)) + lane;
        if (vq == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a) mo[64 * a] = make_float4(E[a + 4].x, E[a + 4].y, O[a + 4].x, O[a + 4].y);
        } else if (vq == 1) {
#pragma unroll
            for (int a = 0; a < 4; ++a) mo[64 * a] = make_float4(E[a + 8].x, E[a + 8].y, O[a + 8].x, O[a + 8].y);
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) mo[64 * a] = make_float4(E[a + 12].x, E[a + 12].y, O[a + 12].x, O[a + 12].y);
        }
    }
}

__global__ __launch_bounds__(64 * kOsWaves) void k_conv_p1_os(const ConvOsArgs A)
{
    ohs_set_fp_mode(A.fp_mode);
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    float4 *cd = reinterpret_cast<float4 *>(smem + kTabComplex);        // [2][16][64] (C, D)
    float2 *tw2 = smem + kTabComplex + kOsCdComplex;                    // [16][64]
    fill_twiddle_tables(tab, A.tw, threadIdx.x, 64 * kOsWaves);
    for (int i = threadIdx.x; i < 2 * kFft; i += 64 * kOsWaves) cd[i] = A.CD[i];
    for (int i = threadIdx.x; i < kFft; i += 64 * kOsWaves) tw2[i] = A.tw2048[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * kOsWaves + wave;
    const long long s64 = gw / A.chunks;
    const int ck = (int)(gw % A.chunks);
    OsWave w;
    w.cd = cd; w.tw2 = tw2; w.tab = tab;
    w.lds = tw2 + kFft + wave * kWaveLdsComplex;
    w.n_frames = (unsigned)A.n_blocks * kBlock;
    w.H = (int)((w.n_frames + kOsHop - 1) / kOsHop);
    w.v_last = w.n_frames - (unsigned)(w.H - 1) * kOsHop;                 // new frames of hop H - 1: 512, 1024 or 1536
    w.last_chunk = ck == A.chunks - 1;
    const int h0 = __builtin_amdgcn_readfirstlane(os_chunk_begin(ck, A.chunks, w.H));
    int h1 = __builtin_amdgcn_readfirstlane(os_chunk_begin(ck + 1, A.chunks, w.H));
    // the state the launch leaves behind needs the 512 outputs BEHIND the last frame: they are in the last window when it
    // has room for them (v_last <= 1024), else one more window of zeros follows
    if (w.last_chunk && w.v_last == kOsHop) ++h1;
    w.h1 = h1;
    const bool active = s64 < (long long)A.n_streams && h0 < h1;
    w.s = active ? (int)s64 : 0;
    w.in_l = A.in + (size_t)w.s * A.in_stream_stride;
    w.in_r = w.in_l + A.in_ch_stride;
    w.out_l = A.out + (size_t)w.s * A.out_stream_stride;
    w.out_r = w.out_l + A.out_ch_stride;
    const PairedPlan plan = paired_plan(lane);

    // frames [1536 h0 - 512, 1536 h0) and the new frames of hop h0: read in front of the barrier -- in place the wave of
    // chunk ck - 1 overwrites the former with its last outputs (in place all chunks of a stream live in this workgroup:
    // launch_conv_p1_os)
    float2 pl[4], pr[4], nl[12], nr[12];
#pragma unroll
    for (int a = 0; a < 4; ++a) { pl[a] = make_float2(0.f, 0.f); pr[a] = make_float2(0.f, 0.f); }
    if (active) {
        if (h0 > 0) {
            const unsigned base = (unsigned)h0 * kOsHop - kBlock + 2u * (unsigned)lane;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                pl[a] = *reinterpret_cast<const float2 *>(w.in_l + base + 128 * a);
                pr[a] = *reinterpret_cast<const float2 *>(w.in_r + base + 128 * a);
            }
        }
        os_load_new<true>(w, h0, lane, nl, nr);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) return;

    // steady hops: whole, not the call's first, and followed by a whole hop OF THE CALL (the steady hop requests the next
    // hop's frames unguarded; at the end of a chunk these belong to the next chunk's wave and are read for nothing)
    const int F = (int)(w.n_frames / kOsHop);                   // whole hops of the call
    const int hs = h1 < F - 1 ? h1 : F - 1;                     // hops [.., hs) are steady
    int h = h0;
    if (h == 0 || h >= hs) { os_hop<true>(A, w, h, lane, plan, pl, pr, nl, nr); ++h; }
    for (; h < hs; ++h) os_hop<false>(A, w, h, lane, plan, pl, pr, nl, nr);
    for (; h < h1; ++h) os_hop<true>(A, w, h, lane, plan, pl, pr, nl, nr);
}

int conv_os_waves_per_cu() { return kOsWaves; }

hipError_t launch_conv_p1_os(const ConvOsArgs &a, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (a.n_blocks <= 0 || a.n_streams <= 0) return hipErrorInvalidValue;
    // In place the wave of hop range ck - 1 overwrites the 512 frames in front of range ck with its last outputs: the
    // ranges of a stream must then share a workgroup (its barrier orders those reads first), so their count must divide the
    // workgroup's waves.  Out of place the ranges are independent and any count goes.
    if (a.chunks < 1) return hipErrorInvalidValue;
    if (a.in == a.out && kOsWaves % a.chunks != 0) return hipErrorInvalidValue;
    if ((unsigned long long)a.n_blocks * kBlock + kOsHop >= (1ull << 31)) return hipErrorInvalidValue;     // 32-bit frame offsets
    if (!a.merged_out || !a.last_in) return hipErrorInvalidValue;       // batch objects only: the lazy state
    // 8-byte accesses: even strides, 8-byte aligned bases
    if ((a.in_stream_stride | a.in_ch_stride | a.out_stream_stride | a.out_ch_stride) & 1) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out)) & 7) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> lds_ok{0};
    const hipError_t e = os_allow_large_lds(reinterpret_cast<const void *>(k_conv_p1_os), kOsLdsBytes, lds_ok);
    if (e != hipSuccess) return e;
    const long long waves = (long long)a.n_streams * a.chunks;
    const dim3 grid((unsigned)((waves + kOsWaves - 1) / kOsWaves));
    if (ev_start || ev_stop) {
        hipExtLaunchKernelGGL(k_conv_p1_os, grid, dim3(64 * kOsWaves), kOsLdsBytes, st, ev_start, ev_stop, 0, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_conv_p1_os, grid, dim3(64 * kOsWaves), kOsLdsBytes, st, a);
    return hipGetLastError();
}

}  // namespace ohs
