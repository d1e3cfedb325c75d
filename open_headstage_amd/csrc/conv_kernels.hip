// conv_kernels.hip -- partitioned FFT convolution kernels for gfx950 (MI355X).
//
// What they compute is fixed by the reference's src/dsp/convolution.rs
// (set_ir :111-139, process_internal_block :184-233, convolve_path_partitioned
// :236-289); how they compute it is MI355X-first:
//   * one wavefront owns one stereo stream and keeps its overlap tails in VGPRs;
//   * L and R are packed into ONE complex FFT-1024 (z = x_l + j x_r) and each
//     speaker's two ear paths into ONE inverse FFT (w = y_*l + j y_*r), so a
//     block costs 1 forward + 2 inverse transforms instead of the reference's 4 + 4;
//   * one spectrum history per stream (packed Z), not one per path; per-path
//     set_ir resets are reproduced by gating partitions on "blocks since reset";
//   * all global traffic is lane-contiguous ([reg][lane] images, 512 B per wave load).
#include "kernels.h"
#include "wave_fft.hpp"
#include "eq_ring2_body.hpp"    // the resident kernel runs the chain's EQ; this file is built -ffp-contract=off too

#include <atomic>

namespace ohs {

// > 64 KiB of dynamic LDS needs an opt-in function attribute, per device.
static hipError_t allow_large_lds(const void *fn, size_t bytes, std::atomic<unsigned long long> &done_mask)
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

// ---------------------------------------------------------------------------------
// IR partition spectra (set_ir, convolution.rs:120-132)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ir_spectrum(const float *__restrict__ ir, int len,
                                                    float2 *__restrict__ dst,
                                                    const float2 *__restrict__ tw)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    float2 *lds = smem + kTabComplex;
    const int lane = threadIdx.x;
    const int part = blockIdx.x;
    fill_twiddle_tables(tab, tw, lane, 64);
    __syncthreads();

    float2 v[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const int idx = part * kBlock + 64 * a + lane;
        v[a] = make_float2((a < 8 && idx < len) ? ir[idx] : 0.0f, 0.0f);
    }
    wave_fft_fwd_zpad(v, lds, tab, lane);
    float2 *o = dst + (size_t)part * kFft;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r * 64 + lane] = v[r];
}

hipError_t launch_ir_spectrum(const float *d_ir, int len, int n_parts, float2 *dst,
                              const float2 *tw, hipStream_t st)
{
    const size_t shmem = (kTabComplex + kWaveLdsComplex) * sizeof(float2);
    hipLaunchKernelGGL(k_ir_spectrum, dim3(n_parts), dim3(64), shmem, st, d_ir, len, dst, tw);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// General convolution kernel: one wave per stream, blocks in sequence.
// ---------------------------------------------------------------------------------
// audio access policies of the general kernel ------------------------------------------------
// PlainIO: planar device buffers (batch mode, multi-launch engine path).
struct PlainIO {
    const float *in_l, *in_r;
    float *out_l, *out_r;
    float gain;
    __device__ __forceinline__ void init(const ConvGeneralArgs &A, size_t s)
    {
        in_l = A.in + s * A.in_stream_stride; in_r = in_l + A.in_ch_stride;
        out_l = A.out + s * A.out_stream_stride; out_r = out_l + A.out_ch_stride;
        gain = A.gain;
    }
    __device__ __forceinline__ float2 load(size_t idx) const { return make_float2(in_l[idx], in_r[idx]); }
    __device__ __forceinline__ void store(size_t idx, float l, float r) const { out_l[idx] = l * gain; out_r[idx] = r * gain; }
};

// EngineIO: the host-block adapter of ConvolutionEngine::process_block (convolution.rs:141-182) folded
// into the kernel.  Input frame i of this call's block stream comes from the
// device FIFO (the < 512 frames left over by earlier calls) or from `src` (this call's n new frames,
// pinned host memory mapped into the device, or the EQ's device output); output frame j goes
// straight to the mapped host buffer when it is popped by this call, else to its post-pop position
// in the device FIFO.  One launch, no separate copies.
// SYSTEM = true (the resident kernel): the host-visible accesses -- this call's input frames and its popped
// output frames, both in pinned host memory -- are system-scope loads / stores (`sc0 sc1`: they bypass this CU's
// L1 and are written through), so that neither a cache invalidate in front of the call nor a write-back of the
// whole L2 behind it is needed; the device-resident state goes through the caches as usual.
template <bool SYSTEM>
struct EngineIOT {
    EngineCallArgs E;
    static __device__ __forceinline__ float ld(const float *p)
    {
        if constexpr (SYSTEM) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else return *p;
    }
    static __device__ __forceinline__ void st(float *p, float v)
    {
        if constexpr (SYSTEM) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else *p = v;
    }
    // Branch-free: the source / destination is chosen by a POINTER select and the access itself is
    // unconditional, so the 16 loads of a block are issued back to back.  (With the `if` around each access
    // hipcc waited for every load before the next branch: eight dependent PCIe round trips, 14 of the 17 us of
    // a 1024-frame call.)
    __device__ __forceinline__ float2 load(size_t idx) const
    {
        const bool old = idx < (size_t)E.in_len;
        const size_t k = idx - (size_t)E.in_len;
        const float *pl = old ? E.in_fifo + idx : E.src_l + k;
        const float *pr = old ? E.in_fifo + E.in_cap + idx : E.src_r + k;
        return make_float2(ld(pl), ld(pr));
    }
    __device__ __forceinline__ void store(size_t idx, float l, float r) const
    {
        const size_t j = (size_t)E.out_len + idx;
        const bool pop = E.emit && j < (size_t)E.n;                    // popped by this call: to the host, with gain
        const size_t f = E.emit ? j - (size_t)E.n : j;                 // else: its post-pop position in the FIFO
        float *pl = pop ? E.host_l + j : E.out_fifo + f;
        float *pr = pop ? E.host_r + j : E.out_fifo + E.out_cap + f;
        const float g = pop ? E.gain_pop : 1.0f;
        st(pl, l * g);
        st(pr, r * g);
    }
};
using EngineIO = EngineIOT<false>;

// ---- the three per-block pieces of convolve_path_partitioned, shared by the sequential general body and
// the one-wave-per-block engine call ----

// (1) inputs of block t -> packed spectrum Z = FFT(x_l + j x_r)  (convolution.rs:245-255)
template <class IO>
__device__ __forceinline__ void block_forward(const IO &io, int t, float2 (&v)[16], float2 *lds, const float2 *tab,
                                              int lane)
{
#pragma unroll
    for (int a = 0; a < 8; ++a) v[a] = io.load((size_t)t * kBlock + 64 * a + lane);
#pragma unroll
    for (int a = 8; a < 16; ++a) v[a] = make_float2(0.0f, 0.0f);   // zero pad (:248-254)
    wave_fft_fwd_zpad(v, lds, tab, lane);
}

// (2) history[history_index] = spectrum (:258); one packed ring per stream
__device__ __forceinline__ void block_store_history(const ConvGeneralArgs &A, float2 *hist, int t,
                                                    const float2 (&v)[16], int lane)
{
    float2 *hs = hist + ((A.cnt + (unsigned long long)t) & (unsigned long long)(A.cap - 1)) * kFft;
#pragma unroll
    for (int r = 0; r < 16; ++r) hs[r * 64 + lane] = v[r];
}

// (3) partitions, newest block with partition 0 (:261-273): acc1 = lsl + j lsr, acc2 = rsl + j rsr spectra
__device__ __forceinline__ void block_mac(const ConvGeneralArgs &A, const float2 *hist, int t, int maxP,
                                          const float2 (&v)[16], float2 (&acc1)[16], float2 (&acc2)[16], int lane)
{
    const unsigned long long mask = (unsigned long long)(A.cap - 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[r] = make_float2(0.f, 0.f); acc2[r] = make_float2(0.f, 0.f); }
    for (int i = 0; i < maxP; ++i) {
        // path p sees block (t - i) only if that block was processed after p's last set_ir
        const bool g0 = i < A.P[0] && (long long)i <= A.since[0] + t;
        const bool g1 = i < A.P[1] && (long long)i <= A.since[1] + t;
        const bool g2 = i < A.P[2] && (long long)i <= A.since[2] + t;
        const bool g3 = i < A.P[3] && (long long)i <= A.since[3] + t;
        if (!(g0 || g1 || g2 || g3)) continue;

        float2 zi[16], zm[16];
        if (i == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) zi[r] = v[r];
        } else {
            const float2 *hs = hist + ((A.cnt + (unsigned long long)t - (unsigned long long)i) & mask) * kFft;
#pragma unroll
            for (int r = 0; r < 16; ++r) zi[r] = hs[r * 64 + lane];
        }
        wave_mirror(zi, zm, lane);
        // Branch-free over the four paths: every spectrum is loaded (partition index clamped into the path's own
        // range) and a gated-off path's value is multiplied by 0, an active one by 1 (exact).  With an `if` around
        // each product hipcc kept every load inside its branch and waited for it there -- 64 dependent L2 round
        // trips, 8.8 of the 16.4 us a 1024-frame call spent in the kernel; a separate straight-line path for the
        // all-active case made it demote the register arrays to scratch memory.
        const float f0 = g0 ? 1.0f : 0.0f, f1 = g1 ? 1.0f : 0.0f, f2 = g2 ? 1.0f : 0.0f, f3 = g3 ? 1.0f : 0.0f;
        const float2 *h0 = A.H[0] + (size_t)(i < A.P[0] ? i : A.P[0] - 1) * kFft;
        const float2 *h1 = A.H[1] + (size_t)(i < A.P[1] ? i : A.P[1] - 1) * kFft;
        const float2 *h2 = A.H[2] + (size_t)(i < A.P[2] ? i : A.P[2] - 1) * kFft;
        const float2 *h3 = A.H[3] + (size_t)(i < A.P[3] ? i : A.P[3] - 1) * kFft;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float2 c0 = h0[r * 64 + lane], c1 = h1[r * 64 + lane], c2 = h2[r * 64 + lane], c3 = h3[r * 64 + lane];
            c0.x *= f0; c0.y *= f0; c1.x *= f1; c1.y *= f1; c2.x *= f2; c2.y *= f2; c3.x *= f3; c3.y *= f3;
            // X_l = (Z + conj(Zm))/2 ; X_r = -j (Z - conj(Zm))/2
            const float2 xl = make_float2(0.5f * (zi[r].x + zm[r].x), 0.5f * (zi[r].y - zm[r].y));
            const float2 xr = make_float2(0.5f * (zi[r].y + zm[r].y), -0.5f * (zi[r].x - zm[r].x));
            cmac(acc1[r], xl, c0);
            cmac_j(acc1[r], xl, c1);
            cmac(acc2[r], xr, c2);
            cmac_j(acc2[r], xr, c3);
        }
    }
}

// One ear pair's share of (3) for the single-stream call, which gives the two halves to two waves: h = 0 -> paths
// 0, 1 (acc = lsl + j lsr), h = 1 -> paths 2, 3 (rsl + j rsr).  Same operations in the same order as block_mac (and
// the same skip rule over all four gates), so the two agree bit for bit.  LDS_P0: partition 0 of path p is read from
// hl[p][1024] in LDS (the resident kernel's copy) instead of A.H[p].
// Partitions i_begin .. i_end - 1 only, on top of `init` (nullptr: zero): the engine kernels' split of the sum into the
// part that involves the call's own blocks and the part computed ahead (ConvGeneralArgs::pre).
template <bool LDS_P0>
__device__ __forceinline__ void block_mac_half(const ConvGeneralArgs &A, const float2 *hist, const float2 *hl, int t,
                                               int i_begin, int i_end, const float2 *init, int h, const float2 (&v)[16],
                                               float2 (&acc)[16], int lane)
{
    const unsigned long long mask = (unsigned long long)(A.cap - 1);
    const int Pa = h ? A.P[2] : A.P[0], Pb = h ? A.P[3] : A.P[1];
    const float2 *Ha = h ? A.H[2] : A.H[0], *Hb = h ? A.H[3] : A.H[1];
    if (init) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = init[r * 64 + lane];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = make_float2(0.f, 0.f);
    }
    for (int i = i_begin; i < i_end; ++i) {
        const bool g0 = i < A.P[0] && (long long)i <= A.since[0] + t;
        const bool g1 = i < A.P[1] && (long long)i <= A.since[1] + t;
        const bool g2 = i < A.P[2] && (long long)i <= A.since[2] + t;
        const bool g3 = i < A.P[3] && (long long)i <= A.since[3] + t;
        if (!(g0 || g1 || g2 || g3)) continue;
        const bool ga = h ? g2 : g0, gb = h ? g3 : g1;

        float2 zi[16], zm[16], ca[16], cb[16];
        if (i == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) zi[r] = v[r];
        } else {
            const float2 *hs = hist + ((A.cnt + (unsigned long long)t - (unsigned long long)i) & mask) * kFft;
#pragma unroll
            for (int r = 0; r < 16; ++r) zi[r] = hs[r * 64 + lane];
        }
        if (LDS_P0 && i == 0) {
            const float2 *la = hl + (2 * h) * kFft, *lb = la + kFft;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ca[r] = la[r * 64 + lane]; cb[r] = lb[r * 64 + lane]; }
        } else {                            // one branch around the whole batch: its 32 loads are in flight together
            const float2 *ha = Ha + (size_t)(i < Pa ? i : Pa - 1) * kFft;
            const float2 *hb = Hb + (size_t)(i < Pb ? i : Pb - 1) * kFft;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ca[r] = ha[r * 64 + lane]; cb[r] = hb[r * 64 + lane]; }
        }
        wave_mirror(zi, zm, lane);
        const float fa = ga ? 1.0f : 0.0f, fb = gb ? 1.0f : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float2 c0 = ca[r], c1 = cb[r];
            c0.x *= fa; c0.y *= fa; c1.x *= fb; c1.y *= fb;
            const float2 xl = make_float2(0.5f * (zi[r].x + zm[r].x), 0.5f * (zi[r].y - zm[r].y));
            const float2 xr = make_float2(0.5f * (zi[r].y + zm[r].y), -0.5f * (zi[r].x - zm[r].x));
            const float2 x = h ? xr : xl;
            cmac(acc[r], x, c0);
            cmac_j(acc[r], x, c1);
        }
    }
}

template <class IO>
__device__ __forceinline__ void conv_general_body(const ConvGeneralArgs &A, const IO &io, float2 *tab, float2 *lds,
                                                  int lane, size_t s)
{
    float2 *hist = A.hist + s * (size_t)A.cap * kFft;
    const float2 *tails = A.tails + s * (size_t)(2 * 8 * 64);
    float2 *tails_out = A.tails_out + s * (size_t)(2 * 8 * 64);

    float2 tail1[8], tail2[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        tail1[a] = A.zero_tails_in ? make_float2(0.f, 0.f) : tails[a * 64 + lane];
        tail2[a] = A.zero_tails_in ? make_float2(0.f, 0.f) : tails[(8 + a) * 64 + lane];
    }
    int maxP = A.P[0];
#pragma unroll
    for (int p = 1; p < 4; ++p) maxP = A.P[p] > maxP ? A.P[p] : maxP;
    const float scale = 1.0f / (float)kFft;   // convolution.rs:280

    for (int t = 0; t < A.n_blocks; ++t) {
        float2 v[16];
        block_forward(io, t, v, lds, tab, lane);
        block_store_history(A, hist, t, v, lane);
        float2 acc1[16], acc2[16];
        block_mac(A, hist, t, maxP, v, acc1, acc2, lane);

        wave_fft_inv(acc1, lds, tab, lane);   // re = lsl, im = lsr   (:276)
        wave_fft_inv(acc2, lds, tab, lane);   // re = rsl, im = rsr

        // overlap-add per path, then ear sums (:279-284, :228-231), then gain (lib.rs:1202-1207)
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const float lsl = fmaf(acc1[a].x, scale, tail1[a].x);
            const float lsr = fmaf(acc1[a].y, scale, tail1[a].y);
            const float rsl = fmaf(acc2[a].x, scale, tail2[a].x);
            const float rsr = fmaf(acc2[a].y, scale, tail2[a].y);
            tail1[a] = make_float2(acc1[a + 8].x * scale, acc1[a + 8].y * scale);
            tail2[a] = make_float2(acc2[a + 8].x * scale, acc2[a + 8].y * scale);
            const size_t idx = (size_t)t * kBlock + 64 * a + lane;
            if (t >= A.skip_store) io.store(idx, lsl + rsl, lsr + rsr);
        }
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        tails_out[a * 64 + lane] = tail1[a];
        tails_out[(8 + a) * 64 + lane] = tail2[a];
    }
}

// (256 VGPRs + 44 AGPRs, one wave per SIMD: the accumulators of both ear pairs, the block's spectrum, its mirror and four
// spectra of a partition are alive together, and hipcc parks some of them in accumulation registers -- register moves, no
// scratch memory.  Since round 4 this kernel is a FALLBACK: long batch calls behind a per-path set_ir go through the
// time-parallel kernels + k_conv_tp_old, real-time sized calls through k_engine_call / k_engine_resident; it serves
// engine calls of a few blocks on the multi-kernel path and batch calls too short for the other plans, and it remains
// the on-GPU cross-check of the fast paths.  Its time is not on any measured path.)
__global__ __launch_bounds__(64) void k_conv_general(const ConvGeneralArgs A)
{
    ohs_set_fp_mode(A.fp_mode);
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    float2 *lds = smem + kTabComplex;
    const int lane = threadIdx.x;
    fill_twiddle_tables(tab, A.tw, lane, 64);
    __syncthreads();
    PlainIO io;
    io.init(A, blockIdx.x);
    conv_general_body(A, io, tab, lds, lane, blockIdx.x);
}

// One ConvolutionEngine::process_block call for a single stream (f2), executed by one workgroup.  One WAVE PER
// BLOCK: the k <= 8 internal blocks of a call only couple through the history ring (P > 1) and the 512-sample
// overlap tails, so they run side by side and exchange those through memory / LDS at two barriers --
// the call's latency is one block's (forward FFT, MAC, two inverse FFTs), not k of them.
// Every thread of the workgroup must call it (it contains workgroup barriers); `tab` holds the twiddle
// tables, `slabs` n_waves per-wave FFT slabs.
// twiddle tables + 8 per-wave FFT slabs + 149 KiB of the CU's 160 either way:
//   k_engine_call     : 8 spectrum hand-over areas + the stored overlaps
//   k_engine_resident : 4 hand-over areas (it serves calls of up to kResidentMaxBlocks blocks, each with a pair of
//                       waves) + the overlaps + partition 0 of the four paths' spectra
constexpr size_t kEngineLdsBytes = (kTabComplex + 8 * (size_t)kWaveLdsComplex + 9 * (size_t)kFft) * sizeof(float2);
// a value the compiler must (re)compute where it is used: keeps loop-invariant per-thread addresses of the resident
// kernel's cold paths from being hoisted out of the call loop, where they would sit in scratch memory across the body
template <class T>
__device__ __forceinline__ T here(T v)
{
    asm volatile("" : "+v"(v));
    return v;
}
#define OHS_DBG_STAMP(i) do { if (dbg && threadIdx.x == 0) dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
// RESIDENT (k_engine_resident): `cache` holds, across calls, the overlaps the engine carries ([16][64], kept up to
// date here) and partition 0 of the four paths' spectra ([4][1024], filled when the kernel starts): the 34 loads
// per wave they replace cost the CU's one address unit ~16 clocks each, next to the PCIe reads everyone waits for.
template <bool SYSTEM, bool RESIDENT>
__device__ __forceinline__ void engine_call_body(const ConvGeneralArgs &A, const EngineCallArgs &E, float2 *tab,
                                                 float2 *slabs, float2 *zx, float2 *cache, int wave, int lane, int n_waves,
                                                 unsigned long long *dbg = nullptr)
{
    OHS_DBG_STAMP(8);
    using IO = EngineIOT<SYSTEM>;
    float2 *lds = slabs + wave * kWaveLdsComplex;
    // 1. frames already waiting in the output FIFO: pop what this call takes, slide the rest down
    if (wave == 0 && E.out_len > 0 && E.emit) {
        const int take = E.out_len < E.n ? E.out_len : E.n;
        for (int i = here(lane); i < take; i += 64) {
            IO::st(E.host_l + i, E.out_fifo[i] * E.gain_pop);
            IO::st(E.host_r + i, E.out_fifo[E.out_cap + i] * E.gain_pop);
        }
        for (int base = E.n; base < E.out_len; base += 64) {      // ascending, read-then-write per 64
            const int i = base + here(lane);
            float l = 0.f, r = 0.f;
            if (i < E.out_len) { l = E.out_fifo[i]; r = E.out_fifo[E.out_cap + i]; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (i < E.out_len) { E.out_fifo[i - E.n] = l; E.out_fifo[E.out_cap + i - E.n] = r; }
        }
    }
    __syncthreads();
    // 2. every complete 512-block (convolution.rs:152-161).  A block's work after the forward transform is two
    // independent halves (paths 0, 1 -> lsl + j lsr; paths 2, 3 -> rsl + j rsr: a MAC and an inverse transform each);
    // with k <= n_waves / 2 blocks in the call every block gets a PAIR of waves, one per half (the even one also
    // runs the forward transform and hands the spectrum over through LDS), otherwise one wave does both halves in
    // turn.  Per (block, half) the results -- overlap for the next block and the raw lower half -- are parked in
    // LDS, and after a barrier the block's waves share the 8 output rows.  Same arithmetic either way.
    IO io;
    io.E = E;
    const bool split = 2 * A.n_blocks <= n_waves;
    const int t = split ? wave >> 1 : wave;
    const int h0 = split ? (wave & 1) : 0, h1 = split ? h0 + 1 : 2;
    const bool active = t < A.n_blocks;
    float2 *hist = A.hist;
    float2 *tails_lds = cache;                      // [16][64]: the overlaps the engine holds from its previous call
    const float2 *hl = cache + kFft;                // RESIDENT: [4][1024] partition 0 of the four paths
    float2 *zt = zx + t * kFft;                     // block t's spectrum, [reg][lane]; later (one wave, both halves) half 0's results
    int maxP = A.P[0];
#pragma unroll
    for (int p = 1; p < 4; ++p) maxP = A.P[p] > maxP ? A.P[p] : maxP;
    const float scale = 1.0f / (float)kFft;         // convolution.rs:280
    // where (block tt, half hh) parks its results: [0, 512) overlap (scaled), [512, 1024) lower half (raw)
    auto parked = [&](int tt, int hh) -> float2 * {
        return split ? slabs + (2 * tt + hh) * kWaveLdsComplex : (hh ? slabs + tt * kWaveLdsComplex : zx + tt * kFft);
    };
    float2 v[16], acc[16];
    const int a0 = split ? 4 * h0 : 0;              // this wave's output rows: a0 .. a0+3 (pair) or 0 .. 7 (alone)
    OHS_DBG_STAMP(9);
    // The CU's address unit takes ~16 clocks per load instruction whatever it fetches, so every load issued now
    // -- by any wave -- stands between the PCIe reads of the input frames and the transform that waits for them
    // (prefetching the spectra here, or keeping the other waves in step with dummy loads, cost 1 - 2.5 us).  Only
    // the waves with a forward transform load, and only their frames.
    const bool fwd = active && h0 == 0;
    if (fwd) {
        block_forward(io, t, v, lds, tab, lane);
        OHS_DBG_STAMP(10);
#pragma unroll
        for (int r = 0; r < 16; ++r) zt[r * 64 + lane] = v[r];
        // with one partition everywhere nothing ever reads the ring: a path only sees blocks processed since
        // its own last set_ir (the `since` gate), and growing a path's partition count IS a set_ir
        if (maxP > 1) block_store_history(A, hist, t, v, lane);
    }
    if constexpr (!RESIDENT) {
        // the engine's stored overlaps (block 0 adds them) into LDS, two rows per wave, so that the output stage
        // finds every block's predecessor in LDS
        const int tr = 2 * wave;                    // n_waves == 8: 16 rows
        const float2 told0 = A.tails[tr * 64 + lane], told1 = A.tails[(tr + 1) * 64 + lane];
        tails_lds[tr * 64 + lane] = told0;
        tails_lds[(tr + 1) * 64 + lane] = told1;
    }
    __threadfence_block();      // the ring slots written above are read by the other waves' partitions
    __syncthreads();            // (also: every input frame has been read before step 3 rewrites the FIFO)
    OHS_DBG_STAMP(11);
    if (active) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = zt[r * 64 + lane];
        // blocks whose sum over the partitions behind the call was computed ahead (behind the previous call: engine_presum in
        // the resident kernel, k_engine_presum behind k_engine_call): only the partitions that pair with the call's own
        // blocks are left
        const bool ahead = t < A.pre_n;
        const int i_end = ahead ? (t + 1 < maxP ? t + 1 : maxP) : maxP;
        for (int h = h0; h < h1; ++h) {
            block_mac_half<RESIDENT>(A, hist, hl, t, 0, i_end, ahead ? A.pre + (size_t)(2 * t + h) * kFft : nullptr, h, v, acc, lane);
            OHS_DBG_STAMP(12);
            wave_fft_inv(acc, lds, tab, lane);      // re = lsl | rsl, im = lsr | rsr   (:276)
            OHS_DBG_STAMP(13);
            float2 *park = (h == h1 - 1) ? lds : zt;
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                park[a * 64 + lane] = make_float2(acc[a + 8].x * scale, acc[a + 8].y * scale);
                park[(8 + a) * 64 + lane] = acc[a];
            }
        }
    }
    __syncthreads();
    if (active) {
        // overlap-add per path, then ear sums (:279-284, :228-231); gain on the popped frames only
        const float2 *lo1 = parked(t, 0) + 512, *lo2 = parked(t, 1) + 512;
        const float2 *pv1 = t ? parked(t - 1, 0) : tails_lds, *pv2 = t ? parked(t - 1, 1) : tails_lds + 512;
        for (int ab = a0; ab < (split ? a0 + 4 : 8); ab += 4) {
            float2 l1[4], l2[4], t1[4], t2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int a = ab + k;
                l1[k] = lo1[a * 64 + lane]; l2[k] = lo2[a * 64 + lane];
                t1[k] = pv1[a * 64 + lane]; t2[k] = pv2[a * 64 + lane];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lsl = fmaf(l1[k].x, scale, t1[k].x), lsr = fmaf(l1[k].y, scale, t1[k].y);
                const float rsl = fmaf(l2[k].x, scale, t2[k].x), rsr = fmaf(l2[k].y, scale, t2[k].y);
                io.store((size_t)t * kBlock + 64 * (ab + k) + lane, lsl + rsl, lsr + rsr);
            }
        }
    }
    OHS_DBG_STAMP(14);
    __syncthreads();            // tails_out may alias tails: block 0 has read the old state by now
    if (active && t == A.n_blocks - 1) {
        for (int h = h0; h < h1; ++h) {
            const float2 *park = parked(t, h);
            const int ln = here(lane);      // (addresses formed here, not hoisted out of the resident kernel's call loop)
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const float2 u = park[a * 64 + ln];
                A.tails_out[(8 * h + a) * 64 + ln] = u;
                if constexpr (RESIDENT) tails_lds[(8 * h + a) * 64 + ln] = u;       // the next call's block 0 reads it here
            }
        }
    }
    // 3. what is left of the input (< 512 frames) becomes the FIFO content for the next call
    if (wave == n_waves - 1) {
        const int consumed = A.n_blocks * kBlock;
        const int rem = E.in_len + E.n - consumed;
        // fewer than 512 frames, i.e. at most 8 per lane and channel: all loads first (clamped index, no branch around
        // a load -- the source is pinned host memory, and one PCIe round trip per loop iteration is what a branchy
        // loop costs), then the stores
        const int n_copy = A.n_blocks == 0 ? E.n : rem;                       // <= 511 either way
        // in_len < 512 <= consumed: the remainder lies entirely in this call's new frames.  Nothing to copy: the
        // (unconditional) loads read frame 0, which every call has.
        const int src_off = (A.n_blocks == 0 || n_copy <= 0) ? 0 : consumed - E.in_len;
        const int dst_off = A.n_blocks == 0 ? E.in_len : 0;
        const int last = n_copy > 0 ? n_copy - 1 : 0;
        float l[8], r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = k * 64 + lane, ic = i < last ? i : last;
            l[k] = IO::ld(E.src_l + src_off + ic);
            r[k] = IO::ld(E.src_r + src_off + ic);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = k * 64 + lane;
            if (i < n_copy) { E.in_fifo[dst_off + i] = l[k]; E.in_fifo[E.in_cap + dst_off + i] = r[k]; }
        }
    }
}

// What the NEXT call's blocks 0 .. n_next - 1 owe to the blocks already in the ring (ConvGeneralArgs::pre): run by every
// wave of the workgroup behind the call's completion word, with `An` = the state AFTER the call (cnt, since advanced).
// Task (j, h) = block j, half h: partitions j + 1 .. maxP - 1 pair with ring slots cnt + j - i < cnt only.  No barrier inside.
// Same products and sums per spectrum value, in the same order, as block_mac_half -- but in no hurry and on few registers
// (the resident kernel's call body leaves none: with block_mac_half inlined a second time hipcc parked loop-invariant
// addresses of the CALL's path in scratch memory): the 16 registers of a spectrum go through in two passes of the
// register groups {g, 3 - g}, which are each other's mirrors (wave_mirror), 8 accumulators at a time.
__device__ __forceinline__ void engine_presum(const ConvGeneralArgs &An, int n_next, int wave_u, int lane, int n_waves)
{
    int maxP = An.P[0];
#pragma unroll
    for (int p = 1; p < 4; ++p) maxP = An.P[p] > maxP ? An.P[p] : maxP;
    const unsigned long long mask = (unsigned long long)(An.cap - 1);
    const int srcA = (63 - lane) << 2, srcB = ((64 - lane) & 63) << 2;
    // A task's two passes are independent (different rows of the spectrum): with few blocks they are dealt to the waves
    // one by one -- a 1 024-frame call's 2 blocks x 2 halves x 2 passes occupy all eight waves --, else a wave runs both.
    // (This loop structure is also the one hipcc allocates without scratch in k_engine_resident; a flat loop over the
    // passes, or block_mac_half itself, cost 1 - 8 spilled registers on the CALL's path: libohs_hip.resources.json.)
    const bool deal_passes = 4 * n_next <= n_waves;
    for (int sub = wave_u; sub < (deal_passes ? 4 : 2) * n_next; sub += n_waves) {
        const int task = deal_passes ? sub >> 1 : sub;
        const int g_begin = deal_passes ? (sub & 1) : 0, g_end = deal_passes ? g_begin + 1 : 2;
        const int j = task >> 1, h = task & 1;
        const int Pa = h ? An.P[2] : An.P[0], Pb = h ? An.P[3] : An.P[1];
        const float2 *Ha = h ? An.H[2] : An.H[0], *Hb = h ? An.H[3] : An.H[1];
        const long long sa = h ? An.since[2] : An.since[0], sb = h ? An.since[3] : An.since[1];
        float2 *dst = An.pre + (size_t)task * kFft;
#pragma unroll 1
        for (int g = g_begin; g < g_end; ++g) {
            // register groups g (rows 4 g ..) and 3 - g; row 4 (g ? 3 : 1) is what lane 0 needs beside them
            const int r1 = 4 * g, r2 = 4 * (3 - g), rx = g ? 12 : 4;
            float2 acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = make_float2(0.f, 0.f);
#pragma unroll 1
            for (int i = j + 1; i < maxP; ++i) {
                const bool ga = i < Pa && (long long)i <= sa + j, gb = i < Pb && (long long)i <= sb + j;
                if (!(ga || gb)) continue;
                const float2 *hs = An.hist + ((An.cnt + (unsigned long long)j - (unsigned long long)i) & mask) * kFft;
                float2 z[8], zm[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { z[k] = hs[(r1 + k) * 64 + lane]; z[4 + k] = hs[(r2 + k) * 64 + lane]; }
                const float2 zx = hs[rx * 64 + lane];
                // wave_mirror, groups g and 3 - g:  zm[4 r + q] <- z[4 (3 - r) + 4 - q] of lane 63 - L (q = 1..3);
                //                                   zm[4 r]     <- z[4 (3 - r)] of lane 64 - L, lane 0: its own z[4 ((4 - r) & 3)]
                auto bp = [](int src, float2 v) {
                    float2 o;
                    o.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.x)));
                    o.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.y)));
                    return o;
                };
#pragma unroll
                for (int q = 1; q < 4; ++q) { zm[q] = bp(srcA, z[4 + 4 - q]); zm[4 + q] = bp(srcA, z[4 - q]); }
                {
                    const float2 m0 = bp(srcB, z[4]), m1 = bp(srcB, z[0]);
                    // own rows: group g = 0: z[4 ((4 - 0) & 3)] = z[0]; group 3: z[4]  (rx);   g = 1: group 1: z[12] (rx); group 2: z[8]
                    const float2 own1 = g ? zx : z[0], own2 = g ? z[4] : zx;
                    zm[0] = lane == 0 ? own1 : m0;
                    zm[4] = lane == 0 ? own2 : m1;
                }
                const float fa = ga ? 1.0f : 0.0f, fb = gb ? 1.0f : 0.0f;
                const float2 *ha = Ha + (size_t)(i < Pa ? i : Pa - 1) * kFft;
                const float2 *hb = Hb + (size_t)(i < Pb ? i : Pb - 1) * kFft;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = (k < 4 ? r1 : r2 - 4) + k;
                    float2 c0 = ha[row * 64 + lane], c1 = hb[row * 64 + lane];
                    c0.x *= fa; c0.y *= fa; c1.x *= fb; c1.y *= fb;
                    const float2 xl = make_float2(0.5f * (z[k].x + zm[k].x), 0.5f * (z[k].y - zm[k].y));
                    const float2 xr = make_float2(0.5f * (z[k].y + zm[k].y), -0.5f * (z[k].x - zm[k].x));
                    const float2 x = h ? xr : xl;
                    cmac(acc[k], x, c0);
                    cmac_j(acc[k], x, c1);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[((k < 4 ? r1 : r2 - 4) + k) * 64 + lane] = acc[k];
        }
    }
}

// One launch = one call.
// `done` (pinned host memory, may be null): the call's sequence number is stored there once every wave's stores have
// drained and been released -- the host spins on that word instead of querying the stream, whose completion signal
// and hipStreamQuery round trip arrive several microseconds after the frames.
__global__ __launch_bounds__(512) void k_engine_call(const ConvGeneralArgs A, const EngineCallArgs E, unsigned *done,
                                                     unsigned seq)
{
    ohs_set_fp_mode(A.fp_mode);
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    fill_twiddle_tables(tab, A.tw, threadIdx.x, blockDim.x);
    float2 *slabs = smem + kTabComplex;
    float2 *zx = slabs + n_waves * kWaveLdsComplex;
    engine_call_body<false, false>(A, E, tab, slabs, zx, zx + n_waves * kFft, wave, lane, n_waves);   // its first barrier also covers the table fill
    if (done) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The launch-per-call form's engine_presum: a kernel of its own, queued right behind k_engine_call (the host launches it
// before it starts waiting for the call's completion word, so the call does not pay for it) -- over the chip instead of
// inside the call's one workgroup: a host that calls back to back (an offline render through the reference's
// process_block loop) waits 5 - 8 us for it instead of 35 - 45.  Workgroup = one wave = (task, spectrum row r): every lane
// owns ONE spectrum value and walks the partitions in engine_presum's order with engine_presum's operations (the value's
// mirror comes from memory instead of a lane permutation: same number), so both forms leave the same bits.
// A = the state AFTER the call.  grid = 2 n_next x 16.
__global__ __launch_bounds__(64) void k_engine_presum(const ConvGeneralArgs A)
{
    ohs_set_fp_mode(A.fp_mode);
    const int lane = threadIdx.x;
    const int task = blockIdx.x >> 4, r = blockIdx.x & 15;
    const int j = task >> 1, h = task & 1;
    int maxP = A.P[0];
#pragma unroll
    for (int p = 1; p < 4; ++p) maxP = A.P[p] > maxP ? A.P[p] : maxP;
    const unsigned long long mask = (unsigned long long)(A.cap - 1);
    const int Pa = h ? A.P[2] : A.P[0], Pb = h ? A.P[3] : A.P[1];
    const float2 *Ha = h ? A.H[2] : A.H[0], *Hb = h ? A.H[3] : A.H[1];
    const long long sa = h ? A.since[2] : A.since[0], sb = h ? A.since[3] : A.since[1];
    // wave_mirror: register 4 rr + q of lane L takes  q = 1..3: register 4 (3 - rr) + 4 - q of lane 63 - L;
    //              q = 0: register 4 (3 - rr) of lane 64 - L, lane 0: its own register 4 ((4 - rr) & 3)
    const int rr = r >> 2, q = r & 3;
    const int rm = q ? 4 * (3 - rr) + 4 - q : (lane == 0 ? 4 * ((4 - rr) & 3) : 4 * (3 - rr));
    const int lm = q ? 63 - lane : (64 - lane) & 63;
    const int own = r * 64 + lane, mir = rm * 64 + lm;
    float2 acc = make_float2(0.f, 0.f);
    constexpr int U = 8;                    // partitions whose four loads are in flight together
    for (int i0 = j + 1; i0 < maxP; i0 += U) {
        float2 z[U], zm[U], ca[U], cb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u < maxP ? i0 + u : maxP - 1;        // (clamped: loaded, not used)
            const float2 *hs = A.hist + ((A.cnt + (unsigned long long)j - (unsigned long long)i) & mask) * kFft;
            z[u] = hs[own]; zm[u] = hs[mir];
            ca[u] = Ha[(size_t)(i < Pa ? i : Pa - 1) * kFft + own];
            cb[u] = Hb[(size_t)(i < Pb ? i : Pb - 1) * kFft + own];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u;
            const bool ga = i < Pa && (long long)i <= sa + j, gb = i < Pb && (long long)i <= sb + j;
            if (i >= maxP || !(ga || gb)) continue;
            const float fa = ga ? 1.0f : 0.0f, fb = gb ? 1.0f : 0.0f;
            float2 c0 = ca[u], c1 = cb[u];
            c0.x *= fa; c0.y *= fa; c1.x *= fb; c1.y *= fb;
            const float2 xl = make_float2(0.5f * (z[u].x + zm[u].x), 0.5f * (z[u].y - zm[u].y));
            const float2 xr = make_float2(0.5f * (z[u].y + zm[u].y), -0.5f * (z[u].x - zm[u].x));
            const float2 x = h ? xr : xl;
            cmac(acc, x, c0);
            cmac_j(acc, x, c1);
        }
    }
    A.pre[(size_t)task * kFft + own] = acc;
}

// RESIDENT form ("persistent kernels + pinned-memory mailboxes"): one workgroup of 8
// waves stays on a CU and serves ConvolutionEngine::process_block calls handed over through a mailbox in
// pinned, mapped host memory -- the host never launches anything per call:
//   start : the kernel copies the engine's state description (pointers, capacities, partition counts, block
//           counters, FIFO fill) from the mailbox into LDS once; from then on it advances the counters itself,
//           with the same integer arithmetic the host mirrors (convolution.rs:149-181)
//   call  : host copies the n input frames to pinned memory, writes {n, gain} and then `seq` into ONE 16-byte
//           word of the mailbox; lane 0 polls that word with a single 16-byte load (one PCIe read per ~1 us)
//           and gets the call's arguments together with its doorbell
//   done  : the workgroup runs engine_call_body, every wave drains its stores, one system-scope release, then
//           `done = seq`; the host spins on it and copies the outputs
// Exit conditions every wave reaches (the decision is taken by one lane and broadcast through LDS in front of
// a barrier): `stop` set by the host (ohs_engine_set_realtime(0), set_ir, clone, destroy, a call the kernel
// cannot serve), or no doorbell for `idle_limit` polls (a host that died or simply went quiet must not leave a
// wave spinning on the GPU); the kernel announces its exit in `alive` so that a host ringing at that moment
// re-launches instead of waiting.
// The EQ is an out-of-line function of the resident kernel: inlined next to the convolution body hipcc needed
// more than the 256 VGPRs two waves per SIMD leave and spilled 350 bytes per lane into the path where every
// microsecond counts (a 1024-frame call without EQ went from 16 to 26 us); as a call it has its own allocation.
// (The body itself stays inlined: out of line it saves and restores ~100 callee-saved registers per call.)
template <class T>
__device__ __forceinline__ void uniform_copy(T &dst, const T &src_lds)
{
    static_assert(sizeof(T) % 4 == 0, "word copy");
    const unsigned *s = reinterpret_cast<const unsigned *>(&src_lds);
    unsigned w[sizeof(T) / 4];
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) w[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)s[i]);
    __builtin_memcpy(&dst, w, sizeof(T));
}

template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p)      // a wave-uniform pointer the compiler can keep in SGPRs
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
// (a function's arguments arrive in VGPRs; the ring body addresses its samples as SGPR base + per-lane offset)
__device__ __attribute__((noinline)) void rt_eq_call(float *buf, long long stride, long long n, int nb,
                                                     const EqPassTable *tab, float *state)
{
    float *ubuf = uniform_ptr(buf);
    const long long ustride = (long long)__builtin_amdgcn_readfirstlane((int)stride);        // (< 2^31: the staging buffer's)
    const long long un = (long long)__builtin_amdgcn_readfirstlane((int)n);
    eq_ring2_wave(ubuf, ubuf, 0, ustride, un, 2, __builtin_amdgcn_readfirstlane(nb), *uniform_ptr(tab), uniform_ptr(state), 0);
}

constexpr unsigned kRtPollBurst = 32;     // pairs of polls per asm statement of the resident kernel's doorbell loop
static_assert(kTabComplex + 8 * kWaveLdsComplex + (kResidentMaxBlocks + 1 + 4) * kFft <= kEngineLdsBytes / sizeof(float2), "LDS plan");
__global__ __launch_bounds__(512) void k_engine_resident(RtMailbox *mb, const float2 *tw, unsigned first_seq,
                                                         unsigned long long idle_limit, unsigned sleep_arg)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    float2 *slabs = smem + kTabComplex;
    __shared__ unsigned s_cmd;
    __shared__ int s_n, s_eq;
    __shared__ unsigned s_eq_ver_call;      // the EQ block version this call wants (low 15 bits, from the call word)
    __shared__ float s_gain;
    __shared__ unsigned s_eq_version;
    __shared__ __attribute__((aligned(16))) unsigned s_eq_blk[(sizeof(RtMailbox) - offsetof(RtMailbox, eq_version) + 3) / 4];
    __shared__ __attribute__((aligned(16))) unsigned s_args[(sizeof(ConvGeneralArgs) + sizeof(EngineCallArgs) + 3) / 4];
    // thread 0's bookkeeping lives in LDS, not in registers: the call body below takes every one of the 256 VGPRs two
    // waves per SIMD leave, and what was alive across it (five 64-bit time stamps, the host pointers) was spilled to
    // scratch memory and reloaded on the call's critical path.
    __shared__ unsigned long long s_st[5];      // doorbell seen, inputs visible, body done, stores drained; shader clock
    __shared__ const float *s_host_src[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    fill_twiddle_tables(tab, tw, threadIdx.x, 512);
    {
        const unsigned *src = reinterpret_cast<const unsigned *>(&mb->conv);
        constexpr int words = (int)((sizeof(ConvGeneralArgs) + sizeof(EngineCallArgs) + 3) / 4);
        for (int i = threadIdx.x; i < words; i += 512) s_args[i] = src[i];
    }
    __syncthreads();
    ConvGeneralArgs &A = *reinterpret_cast<ConvGeneralArgs *>(s_args);
    EngineCallArgs &E = *reinterpret_cast<EngineCallArgs *>(reinterpret_cast<char *>(s_args) + sizeof(ConvGeneralArgs));
    ohs_set_fp_mode(A.fp_mode);             // for the kernel's whole life (a change of mode stops it first), its EQ included
    if (threadIdx.x == 0) s_eq_version = 0xffffffffu;       // no EQ block held yet (the host counts from 1)
    // what the kernel keeps in LDS for as long as it lives (anything that changes it on the host side -- set_ir, a
    // launch-per-call block, clone -- stops the kernel first): the engine's overlaps and partition 0 of every path
    float2 *zx = slabs + 8 * kWaveLdsComplex;
    float2 *cache = zx + kResidentMaxBlocks * kFft;
    for (int i = threadIdx.x; i < kFft; i += 512) cache[i] = A.tails[i];
    for (int p = 0; p < 4; ++p)
        for (int i = threadIdx.x; i < kFft; i += 512) cache[kFft + p * kFft + i] = A.H[p][i];
    __syncthreads();
    if (threadIdx.x == 0) { s_host_src[0] = E.src_l; s_host_src[1] = E.src_r; }
    __syncthreads();
    // CONTROL FLOW OF THIS LOOP IS WAVE-UNIFORM ON PURPOSE.  Every branch that decides whether a wave reaches one of the
    // loop's barriers tests a value the compiler can prove uniform (readfirstlane): hipcc otherwise treats the exit test
    // and the "thread 0 only" blocks as divergent and is free to thread the block that ends one iteration into the one
    // that starts the next -- it then builds an inner loop in which lanes 1..63 of wave 0 and all other waves run ahead
    // through the top barrier while lane 0 is parked, and the call is served again and again without ever being
    // published (seen with the doorbell value kept in LDS; tools/rt_probe.py).  So: the doorbell loop is run by the whole
    // of wave 0 (EXEC is narrowed to lane 0 inside the asm statement, where the compiler cannot see it), its results
    // are broadcast through SGPRs, and lane predication is left to leaf stores.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned last = first_seq;              // the doorbell value already served (meaningful in wave 0)
    for (;;) {
        if (wave_u == 0) {
            unsigned cmd = 0;
            unsigned long long idle = 0;
            for (;;) {
                // {seq, n, gain, stop}: one aligned 16-byte read, so the arguments arrive with their doorbell.
                // TWO reads are kept in flight (a new one is issued while the older one is awaited), so the word is
                // sampled every half PCIe round trip instead of once per round trip: ~0.3 us less from the host's
                // store to the kernel's start.  One asm statement, because a register that is the destination of a
                // load in flight must not be touched by compiler-generated moves; it leaves after kRtPollBurst
                // polls at the latest (idle accounting and the sleep are done outside).  Lane 0 alone issues the reads.
                unsigned w0, w1, w2, w3;
                unsigned sx, sw, budget = kRtPollBurst;
                const unsigned last_s = (unsigned)__builtin_amdgcn_readfirstlane((int)last);
                // (the two destinations are fixed registers, v[8:11] and v[12:15]: inline asm cannot name the
                // components of a 128-bit operand)
                asm volatile(
                    "s_mov_b64 exec, 1\n\t"
                    "global_load_dwordx4 v[8:11], %[p], off sc0 sc1\n"
                    "1:\n\t"
                    "global_load_dwordx4 v[12:15], %[p], off sc0 sc1\n\t"
                    "s_waitcnt vmcnt(1)\n\t"
                    "v_readfirstlane_b32 %[sx], v8\n\t"
                    "v_readfirstlane_b32 %[sw], v11\n\t"
                    "s_nop 3\n\t"
                    "s_cmp_lg_u32 %[sx], %[last]\n\t"
                    "s_cbranch_scc1 2f\n\t"
                    "s_cmp_lg_u32 %[sw], 0\n\t"
                    "s_cbranch_scc1 2f\n\t"
                    "global_load_dwordx4 v[8:11], %[p], off sc0 sc1\n\t"
                    "s_waitcnt vmcnt(1)\n\t"
                    "v_readfirstlane_b32 %[sx], v12\n\t"
                    "v_readfirstlane_b32 %[sw], v15\n\t"
                    "s_nop 3\n\t"
                    "s_cmp_lg_u32 %[sx], %[last]\n\t"
                    "s_cbranch_scc1 3f\n\t"
                    "s_cmp_lg_u32 %[sw], 0\n\t"
                    "s_cbranch_scc1 3f\n\t"
                    "s_sub_u32 %[n], %[n], 1\n\t"
                    "s_cmp_lg_u32 %[n], 0\n\t"
                    "s_cbranch_scc1 1b\n\t"
                    "s_waitcnt vmcnt(0)\n\t"          // budget spent: the younger read (v[8:11]) is complete now
                    "s_branch 2f\n"
                    "3:\n\t"                          // hit in v[12:15]: drain the other read, hand the hit over
                    "s_waitcnt vmcnt(0)\n\t"
                    "v_mov_b32 v8, v12\n\t"
                    "v_mov_b32 v9, v13\n\t"
                    "v_mov_b32 v10, v14\n\t"
                    "v_mov_b32 v11, v15\n"
                    "2:\n\t"                          // result in v[8:11] (lane 0); nothing may stay in flight
                    "s_waitcnt vmcnt(0)\n\t"
                    "v_readfirstlane_b32 %[w0], v8\n\t"
                    "v_readfirstlane_b32 %[w1], v9\n\t"
                    "v_readfirstlane_b32 %[w2], v10\n\t"
                    "v_readfirstlane_b32 %[w3], v11\n\t"
                    "s_mov_b64 exec, -1\n"
                    : [w0] "=&s"(w0), [w1] "=&s"(w1), [w2] "=&s"(w2), [w3] "=&s"(w3), [sx] "=&s"(sx), [sw] "=&s"(sw),
                      [n] "+s"(budget)
                    : [p] "v"(&mb->call_word[0]), [last] "s"(last_s)
                    : "memory", "scc", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
                if (w3) break;                                          // stop
                if (w0 != last) {
                    last = w0; cmd = 1;
                    // w1 = n (bits 0..15) | EQ block version, low 15 bits (16..30) | run the EQ (31)
                    if (lane == 0) {
                        s_n = (int)(w1 & 0xffffu); s_eq = (int)(w1 >> 31); s_eq_ver_call = (w1 >> 16) & 0x7fffu;
                        s_gain = __uint_as_float(w2);
                    }
                    break;
                }
                idle += 2 * kRtPollBurst;
                if (idle >= idle_limit) break;
                for (unsigned k = 0; k < sleep_arg; ++k) __builtin_amdgcn_s_sleep(16);     // ~0.4 us each
            }
            if (lane == 0) s_cmd = cmd;
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane((int)s_cmd) == 0) break;     // (every thread reads s_cmd after the barrier)
        if (threadIdx.x == 0) { s_st[0] = __builtin_amdgcn_s_memrealtime(); s_st[4] = __builtin_amdgcn_s_memtime(); }
        // the host's writes (input frames) precede its doorbell store; make them visible to this CU
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        const int eq_u = __builtin_amdgcn_readfirstlane(s_eq);
        if (eq_u) {
            // lib.rs:1194: the EQ runs on this host block before the engine sees it.  (1) the EQ block (table,
            // state pointer, staging buffer) if the host has changed it, (2) the block's frames from pinned
            // host memory to the device staging buffer, all waves -- the EQ kernel's one-group prefetch cannot
            // cover a PCIe read --, (3) wave 0 = the two chains L, R in place, (4) the convolution reads from there.
            constexpr int eq_words = (int)((sizeof(RtMailbox) - offsetof(RtMailbox, eq_version) + 3) / 4);
            // the version the call wants travels in the call word, so an unchanged block costs no PCIe read
            const unsigned ver_call = (unsigned)__builtin_amdgcn_readfirstlane((int)s_eq_ver_call);
            const unsigned ver_held = (unsigned)__builtin_amdgcn_readfirstlane((int)s_eq_version);
            if (ver_call != (ver_held & 0x7fffu) || ver_held == 0xffffffffu) {
                __syncthreads();
                const unsigned *src = reinterpret_cast<const unsigned *>(&mb->eq_version);
                for (int i = here((int)threadIdx.x); i < eq_words; i += 512) s_eq_blk[i] = src[i];
                __syncthreads();
                if (threadIdx.x == 0) s_eq_version = s_eq_blk[0];
                __syncthreads();
            }
            const RtMailbox *const eqb = reinterpret_cast<const RtMailbox *>(reinterpret_cast<const char *>(s_eq_blk) -
                                                                              offsetof(RtMailbox, eq_version));
            float *const buf = eqb->eq_buf;
            const long long stride = eqb->eq_buf_stride;
            const int n = s_n;
            const int tix = here((int)threadIdx.x);
            // eight PCIe reads per thread in flight (clamped index instead of a branch around each load: one round
            // trip for a 1024-frame block, not one per loop iteration)
            const float *const host_src_l = s_host_src[0], *const host_src_r = s_host_src[1];
            for (int base = 0; base < n; base += 4 * 512) {
                float l[4], r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = base + k * 512 + tix, ic = i < n ? i : n - 1;
                    l[k] = host_src_l[ic]; r[k] = host_src_r[ic];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = base + k * 512 + tix;
                    if (i < n) { buf[i] = l[k]; buf[stride + i] = r[k]; }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (wave_u == 0) {
                if (threadIdx.x == 0) mb->stamps[6] = __builtin_amdgcn_s_memrealtime();
                rt_eq_call(buf, stride, (long long)n, eqb->eq_nb, &eqb->eq_tab, eqb->eq_state);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (threadIdx.x == 0) mb->stamps[7] = __builtin_amdgcn_s_memrealtime();
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {             // this call's FIFO arithmetic (convolution.rs:149-163)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_st[1] = __builtin_amdgcn_s_memrealtime();
            if (eq_u) {
                const RtMailbox *const eqb2 = reinterpret_cast<const RtMailbox *>(reinterpret_cast<const char *>(s_eq_blk) -
                                                                                   offsetof(RtMailbox, eq_version));
                E.src_l = eqb2->eq_buf; E.src_r = eqb2->eq_buf + eqb2->eq_buf_stride;
            } else { E.src_l = s_host_src[0]; E.src_r = s_host_src[1]; }
            const int n = s_n;
            const int k = (E.in_len + n) / kBlock;  // <= kResidentMaxBlocks: the host routes longer calls elsewhere
            A.n_blocks = k;
            E.n = n;
            E.emit = (E.out_len + k * kBlock >= n) ? 1 : 0;
            E.gain_pop = s_gain;
        }
        __syncthreads();
        {
            // the call's descriptors out of LDS into SGPRs (they are uniform, but values read from LDS live in
            // VGPRs, two per pointer, and the body has none to spare)
            ConvGeneralArgs Ar;
            EngineCallArgs Er;
            uniform_copy(Ar, A);
            uniform_copy(Er, E);
            engine_call_body<false, true>(Ar, Er, tab, slabs, zx, cache, wave, lane, 8, mb->stamps);
        }
        if (threadIdx.x == 0) s_st[2] = __builtin_amdgcn_s_memrealtime();
        // publish: every wave's stores (outputs to host memory, state to device memory) drained, then one
        // system-scope release in front of the flag.  (Tried: system-scope `sc0 sc1` loads / stores for the
        // host-visible frames in place of both fences -- engine_call_body<true> -- no faster per call, and not
        // bit-identical to the launch-per-call path in tests/test_gpu_realtime.py; the fences stay.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long st3 = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long st4 = __builtin_amdgcn_s_memrealtime();
            // diagnostics (read by ohs_engine_debug_rt_stamps after the call): doorbell seen, inputs visible,
            // body done, stores drained, released -- written BEFORE the flag, so they are in place when it is seen
            mb->stamps[0] = s_st[0]; mb->stamps[1] = s_st[1]; mb->stamps[2] = s_st[2]; mb->stamps[3] = st3; mb->stamps[4] = st4;
            mb->stamps[5] = __builtin_amdgcn_s_memtime() - s_st[4];     // shader-clock cycles over the same span
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(&mb->done, last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            // advance the counters for the next call
            const int k = A.n_blocks;
            A.cnt += (unsigned long long)k;
            for (int p = 0; p < 4; ++p) A.since[p] += k;
            E.in_len = E.in_len + E.n - k * kBlock;
            E.out_len = E.emit ? E.out_len + k * kBlock - E.n : E.out_len + k * kBlock;
            // the next call's blocks 0 .. k - 1 will find their sums over the partitions behind the call computed (below);
            // a call that completed no block leaves the sums it found valid
            if (k > 0) {
                int mp = A.P[0];
                for (int p = 1; p < 4; ++p) mp = A.P[p] > mp ? A.P[p] : mp;
                A.pre_n = (A.pre != nullptr && mp > 1) ? k : 0;
            }
        }
        __syncthreads();
        {
            // behind the completion word: ConvGeneralArgs::pre for the next call (every wave; no barrier inside; a doorbell
            // that rings meanwhile is seen when wave 0 is back in its loop -- the call then waits for what it would
            // otherwise have computed itself)
            const int k_u = __builtin_amdgcn_readfirstlane(A.n_blocks);
            const int pn_u = __builtin_amdgcn_readfirstlane(A.pre_n);
            if (k_u > 0 && pn_u > 0) {
                ConvGeneralArgs An;
                uniform_copy(An, A);
                engine_presum(An, k_u, wave_u, lane, 8);
                __threadfence_block();
            }
        }
        // (the barrier at the top of the loop orders these updates before the next call reads them)
    }
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(&mb->alive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_engine_resident(RtMailbox *mb_dev, const float2 *tw, unsigned first_seq, unsigned long long idle_limit,
                                  unsigned sleep_arg, hipStream_t st)
{
    const size_t shmem = kEngineLdsBytes;
    static std::atomic<unsigned long long> lds_ok{0};
    const hipError_t e = allow_large_lds(reinterpret_cast<const void *>(k_engine_resident), shmem, lds_ok);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_engine_resident, dim3(1), dim3(512), shmem, st, mb_dev, tw, first_seq, idle_limit, sleep_arg);
    return hipGetLastError();
}

hipError_t launch_conv_general(const ConvGeneralArgs &a, int n_streams, hipStream_t st)
{
    const size_t shmem = (kTabComplex + kWaveLdsComplex) * sizeof(float2);
    hipLaunchKernelGGL(k_conv_general, dim3(n_streams), dim3(64), shmem, st, a);
    return hipGetLastError();
}

hipError_t launch_engine_call(const ConvGeneralArgs &a, const EngineCallArgs &e, hipStream_t st, unsigned *done, unsigned seq)
{
    if (a.n_blocks < 0 || a.n_blocks > 8) return hipErrorInvalidValue;
    // always the full workgroup: up to 4 blocks get a pair of waves each
    static std::atomic<unsigned long long> lds_ok{0};
    const hipError_t e2 = allow_large_lds(reinterpret_cast<const void *>(k_engine_call), kEngineLdsBytes, lds_ok);
    if (e2 != hipSuccess) return e2;
    hipLaunchKernelGGL(k_engine_call, dim3(1), dim3(512), kEngineLdsBytes, st, a, e, done, seq);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess && a.pre_next > 0 && a.pre != nullptr) {
        // the next call's sums over the partitions behind it (ConvGeneralArgs::pre), from the state this call leaves
        ConvGeneralArgs an = a;
        an.cnt += (unsigned long long)a.n_blocks;
        for (int p = 0; p < 4; ++p) an.since[p] += a.n_blocks;
        hipLaunchKernelGGL(k_engine_presum, dim3(2 * a.pre_next * 16), dim3(64), 0, st, an);
        err = hipGetLastError();
    }
    return err;
}

// ---------------------------------------------------------------------------------
// P = 1 fast path (BASELINE configs 2, 3, 5: 4 x 512-tap HRIRs).
//
//   Z  = FFT(x_l + j x_r)                        one forward transform for both channels
//   W  = Z C + conj(Z[N-k]) D                    = Y_l + j Y_r  (all four paths, both ears)
//   w  = IFFT(W):  re = out_l, im = out_r        one inverse transform for both ears
//
// Blocks of one stream only couple through the 512-sample overlap tail, so a stream is cut
// into `chunks` contiguous block ranges that run as independent waves; the tail ENTERING each
// chunk is computed beforehand by the pre-pass (k_conv_p1_edges, below).
// The spectrum history ring is not written: with one partition it is never read
// (convolution.rs:261-273 with P = 1 touches history[idx] only in the block that wrote it).
// The last block of the launch keeps the reference's four separate overlap buffers (two
// packed inverse transforms) so that the state at rest is the general kernel's format and a
// later per-path set_ir can zero exactly one path's overlap (convolution.rs:138).
// ---------------------------------------------------------------------------------
__global__ void k_build_cd(const float2 *__restrict__ h0, const float2 *__restrict__ h1,
                           const float2 *__restrict__ h2, const float2 *__restrict__ h3,
                           float2 *__restrict__ cd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kFft) return;
    const float2 a = make_float2(h0[i].x - h1[i].y, h0[i].y + h1[i].x);   // A = H0 + j H1
    const float2 b = make_float2(h2[i].x - h3[i].y, h2[i].y + h3[i].x);   // B = H2 + j H3
    // The inverse transform's 1/1024 (convolution.rs:279) is folded in here: a power of two, so every later
    // operation of the linear chain scales exactly and the results are bit for bit those of scaling at the end.
    const float h = 0.5f / (float)kFft;
    cd[i] = make_float2(h * (a.x + b.y), h * (a.y - b.x));                // (A - jB)/2 / N
    cd[kFft + i] = make_float2(h * (a.x - b.y), h * (a.y + b.x));         // (A + jB)/2 / N
}
hipError_t launch_build_cd(const float2 *h0, const float2 *h1, const float2 *h2, const float2 *h3,
                           float2 *cd, hipStream_t st)
{
    hipLaunchKernelGGL(k_build_cd, dim3(kFft / 256), dim3(256), 0, st, h0, h1, h2, h3, cd);
    return hipGetLastError();
}

// Launch structure (`out` is written exactly once, in place allowed):
//   1. k_conv_p1_edges -- everything that needs a block's INPUT before the main kernel may overwrite it in place, one
//        transform pair per wave:
//        wave (s, speaker)       : the per-path overlaps the launch's LAST block leaves behind (two paths of one
//                                  speaker packed into one inverse transform), so that the state at rest is the general
//                                  kernel's format and a later per-path set_ir can zero exactly one path's overlap
//                                  (convolution.rs:138)
//        wave (s, e), e = 1..K-1 : the merged overlap leaving block b0(e) - 1, i.e. entering chunk e  -> chunk_tails
//   2. k_conv_p1       -- every block: K chunks per stream as independent waves, one forward + one inverse transform
//                         per block, incoming overlap from chunk_tails (chunk 0: the saved state).
// Round 1 added the chunk-boundary tails afterwards with a read-modify-write pass over `out` (k_conv_p1_fixup);
// recomputing 1 block in ~34 instead costs less than that pass and its traffic.  Round 2 computed the last block's
// output in the pre-pass (with all four overlaps alive: it spilled); since round 3 the main kernel owns every output.
#ifndef OHS_P1_WAVES
#define OHS_P1_WAVES 16
#endif
// The audio of k_conv_p1 is read once and written once: non-temporal loads / stores (-2.7 % on config 3 without EQ,
// three interleaved runs each: 0.583 / 0.577 / 0.587 against 0.605 / 0.595 / 0.594 ms with plain accesses)
#define OHS_P1_LD(p) __builtin_nontemporal_load(p)
#define OHS_P1_ST(p, v) __builtin_nontemporal_store((v), (p))
// waves per workgroup = per CU, four per SIMD.  LDS: 8.5 KiB of twiddle tables + 16 KiB of C / D + 16 slabs of 8672 B
// = 163 840 B, every byte of the CU's 160 KiB (the w64 table hides in the padding of the w1024 table's rows,
// wave_fft.hpp, and a slab ends where its last image does).
constexpr int kP1Waves = OHS_P1_WAVES;
static_assert((kTabComplex + 2 * kFft + kP1Waves * kWaveLdsComplex) * sizeof(float2) <= 160 * 1024, "k_conv_p1: LDS plan");

// W = Z C + conj(Zm) D for the 16 registers of a lane, written over the mirror values (cd: LDS, [2][16][64])
__device__ __forceinline__ void p1_spectral_product(const float2 (&z)[16], float2 (&w)[16], const float2 *cd, int lane)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float2 zz = z[r], m = w[r];
        float2 c, d;        // interleaved in LDS: one ds_read_b128 per register instead of two ds_read_b64
        lds_ld2(reinterpret_cast<const float4 *>(cd) + r * 64 + lane, c, d);
        w[r].x = fmaf(m.y, d.y, fmaf(m.x, d.x, fmaf(-zz.y, c.y, zz.x * c.x)));
        w[r].y = fmaf(-m.y, d.x, fmaf(m.x, d.y, fmaf(zz.y, c.x, zz.x * c.y)));
    }
}

// The same product in the paired bin layout of k_conv_p1 (wave_fft.hpp): the mirror values are the lane's own
// registers; cd holds C and D in that layout ([16][64] float4, filled by the kernel).  Operation for operation the
// arithmetic of p1_spectral_product.
__device__ __forceinline__ void p1_spectral_product_paired(const float2 (&z)[16], float2 (&w)[16], const float2 *cd, int lane)
{
    const bool lane32 = lane == 32;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float2 zz = z[i], m = paired_mirror(z, i >> 2, i & 3, lane32);
        float2 c, d;
        lds_ld2(reinterpret_cast<const float4 *>(cd) + i * 64 + lane, c, d);
        w[i].x = fmaf(m.y, d.y, fmaf(m.x, d.x, fmaf(-zz.y, c.y, zz.x * c.x)));
        w[i].y = fmaf(-m.y, d.x, fmaf(m.x, d.y, fmaf(zz.y, c.x, zz.x * c.y)));
    }
}

// First block of chunk ck of stream s.  Chunk (s, ck) is the work of wave (s * chunks + ck) % 16 of its workgroup,
// and a wave's speed depends on that index: issue arbitration among the four waves of a SIMD is oldest first, so
// waves 0..3 of a workgroup (the oldest of their SIMDs) run 2.3x as fast as waves 12..15 while all sixteen are
// resident -- with equal ranges they finished after 149 / 180 / 217 / 252 us and a quarter of the CU's wave slots
// stood empty until the workgroup (which holds all of the CU's LDS) was done (tools/p1_stamps.py).  Ranges are
// therefore cut in proportion to the age rank's weight w[(wave index) >> 2]; w = {1, 1, 1, 1} is the plain
// equal-length partition.  (Rotating the issue priority per block instead evens the speeds out but lowers the
// SIMD's total rate by a fifth: OHS_P1_PRIO=1.)
// With 2, 4 or 8 chunks per stream (1024 streams and more) consecutive waves would put all chunks of a stream on ONE age
// rank (4 consecutive waves = the four SIMDs' waves of equal age) and the weights could balance nothing: the oldest
// waves' streams would finish 2.3x earlier than the youngest's.  Those launches deal a workgroup's waves out across its
// 16 / K streams instead, so that every stream owns waves of all ranks (K = 2: ranks {0, 3} or {1, 2}, whose weights
// add up alike).
__device__ __forceinline__ bool p1_spread(int chunks)
{
    return kP1Waves == 16 && (chunks == 2 || chunks == 4 || chunks == 8);
}
// wave gw of the main kernel's grid -> (stream, chunk)
__device__ __forceinline__ void p1_wave_job(long long gw, int chunks, long long &s, int &ck)
{
    if (p1_spread(chunks)) {
        const int w = (int)(gw & 15), rank = w >> 2, p = w & 3;
        int sl;
        if (chunks == 4) { sl = p; ck = rank; }
        else if (chunks == 8) { sl = p & 1; ck = 2 * rank + (p >> 1); }
        else { sl = p + ((rank == 1 || rank == 2) ? 4 : 0); ck = rank >> 1; }
        s = (gw >> 4) * (16 / chunks) + sl;
    } else {
        s = gw / chunks;
        ck = (int)(gw % chunks);
    }
}
// age rank of the wave that p1_wave_job gives chunk i of stream s
__device__ __forceinline__ int p1_chunk_rank(int s, int i, int chunks)
{
    if (p1_spread(chunks)) {
        if (chunks == 4) return i;
        if (chunks == 8) return i >> 1;
        return (s & 7) < 4 ? (i ? 3 : 0) : (i ? 2 : 1);
    }
    return (int)(((((unsigned)s * (unsigned)chunks) % (unsigned)kP1Waves + (unsigned)i) % (unsigned)kP1Waves) >> 2) & 3;
}
__device__ __forceinline__ int p1_chunk_begin(int s, int ck, int n_main, int chunks, const int (&w)[4])
{
    static_assert(kP1Waves % 4 == 0 && kP1Waves <= 16, "age ranks: four waves per SIMD");
    if (p1_spread(chunks)) {
        long long c = 0, total = 0;
        for (int i = 0; i < chunks; ++i) {
            const int wi = w[p1_chunk_rank(s, i, chunks)];
            if (i < ck) c += wi;
            total += wi;
        }
        return (int)(c * n_main / total);
    }
    int per_wg = 0;                 // one workgroup's waves: four per age rank (fewer ranks in the OHS_P1_WAVES < 16 experiments)
    for (int i = 0; i < kP1Waves; ++i) per_wg += w[(i >> 2) & 3];
    const unsigned base = ((unsigned)s * (unsigned)chunks) % (unsigned)kP1Waves;
    auto cum = [&](int k) -> long long {
        long long c = (long long)(k / kP1Waves) * per_wg;
        for (int i = 0; i < k % kP1Waves; ++i) c += w[(((base + i) % kP1Waves) >> 2) & 3];
        return c;
    };
    return (int)(cum(ck) * n_main / cum(chunks));
}

#if defined(OHS_EXPERIMENTS) && defined(OHS_P1_STAMPS)
#include "experiments/p1_stamps.inc"        // per-wave life spans and placement (tools/p1_stamps.py)
#else
#define OHS_P1_STAMP_ENTRY()
#define OHS_P1_STAMP_START()
#define OHS_P1_STAMP_END(gw, lane)
#endif
// workgroup index inside the launch's XCD set, or -1 for a workgroup on another XCD (kernels.h: xcd_grid)
__device__ __forceinline__ long long p1_xcd_block(const ConvP1Args &A)
{
    unsigned bid = blockIdx.x;
    if (A.xcd_n != 8) {
        const unsigned x = (bid & 7u) - (unsigned)A.xcd_lo;
        if (x >= (unsigned)A.xcd_n) return -1;
        bid = (bid >> 3) * (unsigned)A.xcd_n + x;
    }
    return (long long)bid;
}

__global__ __launch_bounds__(64 * kP1Waves) void k_conv_p1(const ConvP1Args A)
{
    const long long wg = p1_xcd_block(A);
    if (wg < 0) return;
    ohs_set_fp_mode(A.fp_mode);
    OHS_P1_STAMP_ENTRY();
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    float2 *cd = smem + kTabComplex;                    // [2][16][64]
    fill_twiddle_tables(tab, A.tw, threadIdx.x, 64 * kP1Waves);
    {   // C and D interleaved position by position: cd4[i] = (C[i], D[i]) -- i in the layout the block loop computes in
        float4 *dst = reinterpret_cast<float4 *>(cd);
        for (int i = threadIdx.x; i < kFft; i += 64 * kP1Waves) {
            const int src = paired_to_natural(i & 63, i >> 6);
            const float2 c = A.CD[src], d = A.CD[kFft + src];
            dst[i] = make_float4(c.x, c.y, d.x, d.y);
        }
    }
    __syncthreads();
    // wave-uniform quantities are forced into SGPRs: stream, chunk, block range and the four audio base
    // pointers then cost no VGPRs and the address arithmetic runs on the scalar unit
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2 *lds = smem + kTabComplex + 2 * kFft + wave * kWaveLdsComplex;
    const long long gw = wg * kP1Waves + wave;
    long long s64;
    int ck;
    p1_wave_job(gw, A.chunks, s64, ck);
    // A.own_tails: the wave computes the overlap entering its chunk itself -- one dry block in front of its range (b0 - 1,
    // nothing stored) instead of a boundary tail from the pre-pass.  In place that block's input is the LAST thing the
    // wave of chunk ck - 1 overwrites, so every wave of the workgroup (all chunks of a stream live in one workgroup in
    // these launches: launch_conv_p1) reads its first inputs in front of a barrier and stores behind it; waves without
    // work take part in the barrier too.
    bool active = s64 < (long long)A.n_streams;
    const int s = active ? (int)s64 : 0;
    const int n_main = A.n_blocks;        // every block's output is this kernel's
    const int b0 = __builtin_amdgcn_readfirstlane(p1_chunk_begin(s, ck, n_main, A.chunks, A.weights));
    const int b1 = __builtin_amdgcn_readfirstlane(p1_chunk_begin(s, ck + 1, n_main, A.chunks, A.weights));
    active = active && b0 < b1;
    if (!active) {
        if (A.own_tails) __syncthreads();
        return;
    }
    const bool own_tail = A.own_tails && ck > 0;
    const int t_first = own_tail ? b0 - 1 : b0;
    OHS_P1_STAMP_START();

    const float *in_l = A.in + (size_t)s * A.in_stream_stride;
    const float *in_r = in_l + A.in_ch_stride;
    float *out_l = A.out + (size_t)s * A.out_stream_stride;
    float *out_r = out_l + A.out_ch_stride;

    float2 tail[8];
    if (ck == 0 && A.merged_in) {
        const float2 *mt = A.merged_in + (size_t)s * (8 * 64);
#pragma unroll
        for (int a = 0; a < 8; ++a) tail[a] = mt[a * 64 + lane];
    } else if (ck == 0) {
        const float2 *tails = A.tails + (size_t)s * (2 * 8 * 64);
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const float2 t1 = tails[a * 64 + lane], t2 = tails[(8 + a) * 64 + lane];
            tail[a] = make_float2(t1.x + t2.x, t1.y + t2.y);
        }
    } else if (own_tail) {
#pragma unroll
        for (int a = 0; a < 8; ++a) tail[a] = make_float2(0.0f, 0.0f);
    } else {
        const float2 *ct = A.chunk_tails + ((size_t)s * A.chunks + ck) * (size_t)(8 * 64);
#pragma unroll
        for (int a = 0; a < 8; ++a) tail[a] = ct[a * 64 + lane];
    }
    // All waves of a workgroup leave the table barrier together and run the same program, so their LDS
    // bursts and their arithmetic phases coincide; a start offset per wave spreads them over the block time.
    for (int i = 0; i < wave * A.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    // The next block's 16 input dwords are requested BEFORE this block's 16 stores are issued: vector-memory
    // operations retire in order, so loads issued after the stores (at the top of the next iteration) could not be
    // consumed before every one of those stores had been acknowledged.  (Requesting them a whole inverse
    // transform earlier would hide the HBM latency entirely, but needs 16 more VGPRs at the kernel's register
    // peak: hipcc spilled 80 VGPRs at 4 waves per SIMD and still 24 at 3.)
    float xl[8], xr[8];
    // one address per channel and direction, advanced by a block per iteration; every access is base + immediate
    const float *pl = in_l + (size_t)t_first * kBlock + lane, *pr = in_r + (size_t)t_first * kBlock + lane;
    float *ql = out_l + (size_t)t_first * kBlock + lane, *qr = out_r + (size_t)t_first * kBlock + lane;
#pragma unroll
    for (int a = 0; a < 8; ++a) { xl[a] = OHS_P1_LD(&pl[64 * a]); xr[a] = OHS_P1_LD(&pr[64 * a]); }
    if (A.own_tails) __syncthreads();       // (waits for the loads above: no wave has stored anything yet)
    // Issue arbitration between the four waves of a SIMD is "priority, then age": left alone, the oldest wave of
    // every SIMD runs almost unimpeded and the youngest gets the leftover slots -- waves 0..3 of a workgroup finished
    // their ranges after 156 us, waves 12..15 after 268 us (tools/p1_stamps.py), and a CU's slots stood empty for a
    // quarter of the launch, because the workgroup holds all of the CU's LDS until its last wave is done.  So the
    // priority rotates: in block k the wave of age rank g (wave >> 2) runs at priority (g + k) & 3, every wave gets
    // every level a quarter of the time and all sixteen finish together.
    const int age_rank = wave >> 2;
    int prio_phase = age_rank;
    const PairedPlan plan = paired_plan(lane);
    for (int t = t_first; t < b1; ++t) {
        const bool dry = t < b0;            // the block in front of the range: its overlap is all that is wanted
        if (A.prio_mode == 1) {
            switch (prio_phase & 3) {       // (s_setprio takes an immediate)
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
            ++prio_phase;
        }
        if (A.last_in && t == A.n_blocks - 1) {     // lazy state: the launch's last block keeps a copy of its input
            float *li = A.last_in + (size_t)s * (2 * kBlock) + lane;
#pragma unroll
            for (int a = 0; a < 8; ++a) { li[64 * a] = xl[a]; li[kBlock + 64 * a] = xr[a]; }
        }
        float2 v[16];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            v[a] = make_float2(xl[a], xr[a]);
            v[a + 8] = make_float2(0.0f, 0.0f);
        }
        float2 w[16];
        wave_fft_fwd_paired(v, lds, tab, lane, plan);
        p1_spectral_product_paired(v, w, cd, lane);
        // the prefetch of block t + 1 needs no guard except behind the launch's last block, where it re-reads that
        // block (its own frames: in bounds; the values are never used)
        const int adv = (t + 1 < A.n_blocks) ? kBlock : 0;
        pl += adv; pr += adv;
        wave_fft_inv_paired(w, lds, tab, lane, plan);
#pragma unroll
        for (int a = 0; a < 8; ++a) { xl[a] = OHS_P1_LD(&pl[64 * a]); xr[a] = OHS_P1_LD(&pr[64 * a]); }
        if (!dry) {
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                OHS_P1_ST(&ql[64 * a], (w[a].x + tail[a].x) * A.gain);         // (1/N is in C and D)
                OHS_P1_ST(&qr[64 * a], (w[a].y + tail[a].y) * A.gain);
            }
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) tail[a] = w[a + 8];
        ql += kBlock; qr += kBlock;
    }
    if (A.merged_out && b1 == A.n_blocks) {         // lazy state: the merged overlap the launch leaves behind
        float2 *mo = A.merged_out + (size_t)s * (8 * 64);
#pragma unroll
        for (int a = 0; a < 8; ++a) mo[a * 64 + lane] = tail[a];
    }
    OHS_P1_STAMP_END(gw, lane);
}

// pre-pass, two small kernels that read INPUT frames only -- which is why they run before the main kernel stores over
// them (in place) -- one transform pair per wave:
//   k_conv_p1_state  wave (s, speaker): the overlap the launch's last block leaves behind in the reference's per-path
//                    form (convolution.rs:279-284), one speaker's two paths per wave: forward transform, that speaker's
//                    spectrum times (H_a + j H_b), inverse, upper half -> tails_out.  4-wave workgroups (its 64 spectrum
//                    registers on top of the transform's do not fit the 128 a 16-wave workgroup allows).
//   k_conv_p1_edges  wave (s, e >= 1): the merged overlap leaving block b0(e) - 1 -> chunk_tails; same workgroup
//                    shape as the main kernel (the 25 KiB of tables are fetched once per 16 waves).
// Round 2 did the last block's output here as well, with the incoming overlap, the spectrum, its mirror and both
// per-speaker spectra alive together: 128 VGPRs + 244 B of scratch per lane.  The output of every block now belongs to
// the main kernel, and neither pre-pass kernel spills.
constexpr int kP1StateWaves = 4;
__global__ __launch_bounds__(64 * kP1StateWaves) void k_conv_p1_state(const ConvP1Args A)
{
    ohs_set_fp_mode(A.fp_mode);
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    fill_twiddle_tables(tab, A.tw, threadIdx.x, 64 * kP1StateWaves);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2 *lds = smem + kTabComplex + wave * kWaveLdsComplex;
    const long long gw = (long long)blockIdx.x * kP1StateWaves + wave;
    if (gw >= 2ll * A.n_streams) return;
    const int s = (int)(gw >> 1), spk = (int)(gw & 1);
    const float *in_l = A.in + (size_t)s * A.in_stream_stride;
    const float *in_r = in_l + A.in_ch_stride;
    const unsigned base = (unsigned)(A.n_blocks - 1) * kBlock + lane;
    float2 v[16], w[16];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        v[a] = make_float2(in_l[base + 64 * a], in_r[base + 64 * a]);
        v[a + 8] = make_float2(0.0f, 0.0f);
    }
    wave_fft_fwd_mirror(v, w, lds, tab, lane);
    const float2 *ha = A.H[2 * spk], *hb = A.H[2 * spk + 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float2 zm = w[r];
        // X_l = (Z + conj Zm) / 2, X_r = (Z - conj Zm) / 2j
        const float2 x = spk == 0 ? make_float2(0.5f * (v[r].x + zm.x), 0.5f * (v[r].y - zm.y))
                                  : make_float2(0.5f * (v[r].y + zm.y), -0.5f * (v[r].x - zm.x));
        float2 y = make_float2(0.0f, 0.0f);
        cmac(y, x, ha[r * 64 + lane]);
        cmac_j(y, x, hb[r * 64 + lane]);
        w[r] = y;
    }
    wave_fft_inv(w, lds, tab, lane);
    const float scale = 1.0f / (float)kFft;
    float2 *tails_out = A.tails_out + (size_t)s * (2 * 8 * 64) + (size_t)spk * (8 * 64);
#pragma unroll
    for (int a = 0; a < 8; ++a) tails_out[a * 64 + lane] = make_float2(w[a + 8].x * scale, w[a + 8].y * scale);
}

__global__ __launch_bounds__(64 * kP1Waves) void k_conv_p1_edges(const ConvP1Args A)
{
    const long long wg = p1_xcd_block(A);
    if (wg < 0) return;
    ohs_set_fp_mode(A.fp_mode);
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    float2 *cd = smem + kTabComplex;
    fill_twiddle_tables(tab, A.tw, threadIdx.x, 64 * kP1Waves);
    {   // C and D interleaved position by position: cd4[i] = (C[i], D[i])
        float4 *dst = reinterpret_cast<float4 *>(cd);
        for (int i = threadIdx.x; i < kFft; i += 64 * kP1Waves) {
            const float2 c = A.CD[i], d = A.CD[kFft + i];
            dst[i] = make_float4(c.x, c.y, d.x, d.y);
        }
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2 *lds = smem + kTabComplex + 2 * kFft + wave * kWaveLdsComplex;
    const long long gw = wg * kP1Waves + wave;
    if (gw >= (long long)A.n_streams * (A.chunks - 1)) return;
    const int s = (int)(gw / (A.chunks - 1));
    const int e = 1 + (int)(gw % (A.chunks - 1));
    const float *in_l = A.in + (size_t)s * A.in_stream_stride;
    const float *in_r = in_l + A.in_ch_stride;
    // host (conv_launch): chunks <= n_blocks, and weights other than {1,1,1,1} only when even a stream whose chunks all
    // sit on the heaviest ranks keeps >= 2 blocks in its shortest chunk (n_blocks * wmin >= 2 * chunks * wmax): b0 >= 1
    const int b0 = __builtin_amdgcn_readfirstlane(p1_chunk_begin(s, e, A.n_blocks, A.chunks, A.weights));
    const unsigned base = (unsigned)(b0 - 1) * kBlock + lane;
    float2 v[16], w[16];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        v[a] = make_float2(in_l[base + 64 * a], in_r[base + 64 * a]);
        v[a + 8] = make_float2(0.0f, 0.0f);
    }
    wave_fft_fwd_mirror(v, w, lds, tab, lane);
    p1_spectral_product(v, w, cd, lane);
    wave_fft_inv(w, lds, tab, lane);
    float2 *ct = A.chunk_tails + ((size_t)s * A.chunks + e) * (size_t)(8 * 64);
#pragma unroll
    for (int a = 0; a < 8; ++a) ct[a * 64 + lane] = w[a + 8];        // (1/N is in C and D)
}

int conv_p1_waves_per_cu() { return kP1Waves; }

hipError_t launch_conv_p1_state(const ConvP1Args &a, hipStream_t st)
{
    if (a.n_blocks <= 0 || a.n_streams <= 0 || !a.tails_out) return hipErrorInvalidValue;
    const size_t shmem_state = (kTabComplex + kP1StateWaves * kWaveLdsComplex) * sizeof(float2);
    const long long waves = 2ll * a.n_streams;
    hipLaunchKernelGGL(k_conv_p1_state, dim3((unsigned)((waves + kP1StateWaves - 1) / kP1StateWaves)),
                       dim3(64 * kP1StateWaves), shmem_state, st, a);
    return hipGetLastError();
}

hipError_t launch_conv_p1(const ConvP1Args &a, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (a.n_blocks <= 0 || a.n_streams <= 0 || a.chunks < 1) return hipErrorInvalidValue;
    if (a.chunks > a.n_blocks) return hipErrorInvalidValue;
    if ((unsigned long long)a.n_blocks * kBlock >= (1ull << 32)) return hipErrorInvalidValue;   // 32-bit frame offsets
    if (a.xcd_n < 1 || a.xcd_n > 8 || a.xcd_lo < 0 || a.xcd_lo + a.xcd_n > 8) return hipErrorInvalidValue;
    const size_t shmem_main = (kTabComplex + 2 * kFft + kP1Waves * kWaveLdsComplex) * sizeof(float2);
    if ((a.merged_out == nullptr) != (a.last_in == nullptr)) return hipErrorInvalidValue;
    const bool single_kernel = a.merged_out && (a.own_tails || a.chunks == 1);
    if (ev_start && !single_kernel) { hipError_t e = hipEventRecord(ev_start, st); if (e != hipSuccess) return e; }
    if (!a.merged_out) {   // 1a. the state the launch leaves behind, in the per-path form (else: lazily, kernels.h)
        hipError_t e = launch_conv_p1_state(a, st);
        if (e != hipSuccess) return e;
    }
    if (a.own_tails && !(kP1Waves == 16 && (a.chunks == 2 || a.chunks == 4 || a.chunks == 8 || a.chunks == 16)))
        return hipErrorInvalidValue;        // all chunks of a stream must live in one workgroup (p1_wave_job)
    if (a.chunks > 1 && !a.own_tails) {   // 1b. the boundary tails
        static std::atomic<unsigned long long> lds_ok{0};
        hipError_t e = allow_large_lds(reinterpret_cast<const void *>(k_conv_p1_edges), shmem_main, lds_ok);
        if (e != hipSuccess) return e;
        const long long waves = (long long)a.n_streams * (a.chunks - 1);
        hipLaunchKernelGGL(k_conv_p1_edges, dim3(xcd_grid((unsigned)((waves + kP1Waves - 1) / kP1Waves), a.xcd_n)),
                           dim3(64 * kP1Waves), shmem_main, st, a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    {       // 2. every block: `chunks` independent waves per stream
        const long long waves = (long long)a.n_streams * a.chunks;
        static std::atomic<unsigned long long> lds_ok{0};
        hipError_t e = allow_large_lds(reinterpret_cast<const void *>(k_conv_p1), shmem_main, lds_ok);
        if (e != hipSuccess) return e;
        const dim3 grid(xcd_grid((unsigned)((waves + kP1Waves - 1) / kP1Waves), a.xcd_n));
        if (single_kernel && (ev_start || ev_stop)) {
            hipExtLaunchKernelGGL(k_conv_p1, grid, dim3(64 * kP1Waves), shmem_main, st, ev_start, ev_stop, 0, a);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(k_conv_p1, grid, dim3(64 * kP1Waves), shmem_main, st, a);
        e = hipGetLastError();
        if (e == hipSuccess && ev_stop) e = hipEventRecord(ev_stop, st);
        return e;
    }
}

// ---------------------------------------------------------------------------------
// Time-parallel long-IR path (BASELINE config 4: 4 x 16384 taps, P = 32).
//
// convolve_path_partitioned (convolution.rs:236-289) is a P-tap FIR along the BLOCK axis, per
// frequency bin:  W_t[k] = sum_i ( Z_{t-i}[k] C_i[k] + conj(Z_{t-i}[N-k]) D_i[k] ).  Nothing in
// it is sequential, so the three stages run over all blocks of a segment at once, and the MAC
// keeps a (time x partition) tile in registers so that each history spectrum is fetched
// ~(TB + 7)/TB^2 times per use instead of once per use: the 2(P-1) x 4104 B per block that a
// block-at-a-time engine streams from HBM become L2 hits on an 8 KiB-per-block ring.
// ---------------------------------------------------------------------------------
__global__ void k_build_cd_multi(const float2 *__restrict__ h0, const float2 *__restrict__ h1,
                                 const float2 *__restrict__ h2, const float2 *__restrict__ h3,
                                 int p0, int p1, int p2, int p3, int Ppad, float2 *__restrict__ cd)
{
    const int i = blockIdx.y;                                  // partition
    const int b = blockIdx.x * blockDim.x + threadIdx.x;       // layout position
    if (i >= Ppad || b >= kFft) return;
    const float2 z = make_float2(0.f, 0.f);
    const float2 a0 = i < p0 ? h0[(size_t)i * kFft + b] : z;
    const float2 a1 = i < p1 ? h1[(size_t)i * kFft + b] : z;
    const float2 a2 = i < p2 ? h2[(size_t)i * kFft + b] : z;
    const float2 a3 = i < p3 ? h3[(size_t)i * kFft + b] : z;
    const float2 A_ = make_float2(a0.x - a1.y, a0.y + a1.x);
    const float2 B_ = make_float2(a2.x - a3.y, a2.y + a3.x);
    cd[((size_t)i * 2 + 0) * kFft + b] = make_float2(0.5f * (A_.x + B_.y), 0.5f * (A_.y - B_.x));
    cd[((size_t)i * 2 + 1) * kFft + b] = make_float2(0.5f * (A_.x - B_.y), 0.5f * (A_.y + B_.x));
}
hipError_t launch_build_cd_multi(const float2 *h0, const float2 *h1, const float2 *h2, const float2 *h3,
                                 int p0, int p1, int p2, int p3, int Ppad, float2 *cd, hipStream_t st)
{
    hipLaunchKernelGGL(k_build_cd_multi, dim3(kFft / 256, Ppad), dim3(256), 0, st, h0, h1, h2, h3, p0, p1,
                       p2, p3, Ppad, cd);
    return hipGetLastError();
}

// F: one wave per (stream, block).  kTpWaves waves per workgroup (one workgroup per CU): the 8.5 KiB of twiddle
// tables are fetched once per 16 waves instead of once per 4, and a CU holds 16 waves instead of 12.
#ifndef OHS_TP_WAVES
#define OHS_TP_WAVES 16
#endif
constexpr int kTpWaves = OHS_TP_WAVES;      // LDS: 8.5 KiB + 16 x 8672 B = 144 KiB
__global__ __launch_bounds__(64 * kTpWaves) void k_conv_tp_forward(const ConvTpArgs A)
{
    ohs_set_fp_mode(A.fp_mode);
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    fill_twiddle_tables(tab, A.tw, threadIdx.x, 64 * kTpWaves);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2 *lds = smem + kTabComplex + wave * kWaveLdsComplex;
    const long long gw = (long long)blockIdx.x * kTpWaves + wave;
    if (gw >= (long long)A.n_streams * A.seg_blocks) return;
    const size_t s = (size_t)(gw / A.seg_blocks);
    const int t = (int)(gw % A.seg_blocks);
    const float *in_l = A.in + s * A.in_stream_stride;
    const float *in_r = in_l + A.in_ch_stride;
    float2 v[16];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const size_t idx = (size_t)t * kBlock + 64 * a + lane;
        v[a] = make_float2(in_l[idx], in_r[idx]);
        v[a + 8] = make_float2(0.f, 0.f);
    }
    if (A.xh_ring) {            // the block's frames join the input history (kernels.h: ConvTpArgs::xh_ring); wave-uniform
        const long long f0 = A.xh_frame0 + (long long)t * kBlock;
        if (f0 >= 0) {
            const long long L2 = 2 * A.xh_len;
            float *xl = A.xh_ring + s * 2 * L2 + (A.xh_head + f0) % L2, *xr = xl + L2;
#pragma unroll
            for (int a = 0; a < 8; ++a) { xl[64 * a + lane] = v[a].x; xr[64 * a + lane] = v[a].y; }
        }
    }
    wave_fft_fwd_zpad(v, lds, tab, lane);
    float2 *hs = A.hist + (s * (size_t)A.cap + (size_t)((A.cnt + (unsigned long long)t) & (unsigned long long)(A.cap - 1))) * kFft;
#pragma unroll
    for (int r = 0; r < 16; ++r) hs[r * 64 + lane] = v[r];
}
hipError_t launch_conv_tp_forward(const ConvTpArgs &a, hipStream_t st)
{
    const size_t shmem = (kTabComplex + kTpWaves * kWaveLdsComplex) * sizeof(float2);
    static std::atomic<unsigned long long> lds_ok{0};
    const hipError_t e = allow_large_lds(reinterpret_cast<const void *>(k_conv_tp_forward), shmem, lds_ok);
    if (e != hipSuccess) return e;
    const long long waves = (long long)a.n_streams * a.seg_blocks;
    hipLaunchKernelGGL(k_conv_tp_forward, dim3((unsigned)((waves + kTpWaves - 1) / kTpWaves)), dim3(64 * kTpWaves), shmem, st, a);
    return hipGetLastError();
}

// I: one wave per (stream, block < n_mac); the 2 S trailing waves of the grid turn the last block's spectra into the
// new state: overlaps (lsl + j lsr) from W1 and (rsl + j rsr) from W - W1 (until round 3 a kernel of their own)
__global__ __launch_bounds__(64 * kTpWaves) void k_conv_tp_inverse(const ConvTpArgs A)
{
    ohs_set_fp_mode(A.fp_mode);
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    fill_twiddle_tables(tab, A.tw, threadIdx.x, 64 * kTpWaves);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2 *lds = smem + kTabComplex + wave * kWaveLdsComplex;
    const long long gw = (long long)blockIdx.x * kTpWaves + wave;
    const long long n_main = (long long)A.n_streams * A.n_mac;
    if (gw >= n_main + 2ll * A.n_streams) return;
    const float scale = 1.0f / (float)kFft;
    float2 v[16];
    if (gw >= n_main) {
        const size_t s = (size_t)((gw - n_main) >> 1);
        const int which = (int)((gw - n_main) & 1);
        const float2 *Wl = A.W + (s * (size_t)A.n_mac + (size_t)(A.n_mac - 1)) * kFft;
        const float2 *W1 = A.W1 + s * (size_t)kFft;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float2 w1 = W1[r * 64 + lane];
            if (which == 0) v[r] = w1;
            else { const float2 w = Wl[r * 64 + lane]; v[r] = make_float2(w.x - w1.x, w.y - w1.y); }
        }
        wave_fft_inv(v, lds, tab, lane);
        float2 *to = A.tails_out + s * (size_t)(2 * 8 * 64) + (size_t)which * (8 * 64);
#pragma unroll
        for (int a = 0; a < 8; ++a) to[a * 64 + lane] = make_float2(v[a + 8].x * scale, v[a + 8].y * scale);
        return;
    }
    const size_t s = (size_t)(gw / A.n_mac);
    const int t = (int)(gw % A.n_mac);
    const float2 *Wt = A.W + (s * (size_t)A.n_mac + (size_t)t) * kFft;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = Wt[r * 64 + lane];
    if (t > 0) {
        // fold the previous block's overlap in: IFFT((-1)^k W_{t-1})[n] = w_{t-1}[n + 512];
        // bin k = 4 lane + (r & 3) + 256 (r >> 2) is odd exactly when r is odd
        const float2 *Wp = Wt - kFft;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float2 p = Wp[r * 64 + lane];
            if (r & 1) { v[r].x -= p.x; v[r].y -= p.y; } else { v[r].x += p.x; v[r].y += p.y; }
        }
    }
    wave_fft_inv(v, lds, tab, lane);
    float *out_l = A.out + s * A.out_stream_stride;
    float *out_r = out_l + A.out_ch_stride;
    const float2 *tails = A.tails + s * (size_t)(2 * 8 * 64);
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        float ol = v[a].x * scale, orr = v[a].y * scale;
        if (t == 0) {       // first block of the segment: overlap left by the previous call
            const float2 t1 = tails[a * 64 + lane], t2 = tails[(8 + a) * 64 + lane];
            ol += t1.x + t2.x;
            orr += t1.y + t2.y;
        }
        const size_t idx = (size_t)t * kBlock + 64 * a + lane;
        out_l[idx] = ol * A.gain;
        out_r[idx] = orr * A.gain;
    }
}

hipError_t launch_conv_tp_inverse(const ConvTpArgs &a, hipStream_t st)
{
    if (a.n_mac <= 0) return hipSuccess;
    const size_t shmem = (kTabComplex + kTpWaves * kWaveLdsComplex) * sizeof(float2);
    static std::atomic<unsigned long long> lds_ok{0};
    const hipError_t e = allow_large_lds(reinterpret_cast<const void *>(k_conv_tp_inverse), shmem, lds_ok);
    if (e != hipSuccess) return e;
    const long long waves = (long long)a.n_streams * a.n_mac + 2ll * a.n_streams;     // + the new state, two waves per stream
    hipLaunchKernelGGL(k_conv_tp_inverse, dim3((unsigned)((waves + kTpWaves - 1) / kTpWaves)), dim3(64 * kTpWaves), shmem, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// Diagnostic: natural-order complex FFT-1024 through wave_fft_fwd / wave_fft_inv
// (one wave per transform).  Used by the tests to measure the primitive's accuracy.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_debug_fft(const float2 *__restrict__ in, float2 *__restrict__ out,
                                                  int inverse, const float2 *__restrict__ tw)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tab = smem;
    float2 *lds = smem + kTabComplex;
    const int lane = threadIdx.x;
    fill_twiddle_tables(tab, tw, lane, 64);
    __syncthreads();
    const float2 *src = in + (size_t)blockIdx.x * kFft;
    float2 *dst = out + (size_t)blockIdx.x * kFft;
    float2 v[16];
    if (!inverse) {
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = src[64 * a + lane];
        wave_fft_fwd(v, lds, tab, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[4 * lane + j + 256 * r] = v[4 * r + j];
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * r + j] = src[4 * lane + j + 256 * r];
        wave_fft_inv(v, lds, tab, lane);
#pragma unroll
        for (int a = 0; a < 16; ++a) dst[64 * a + lane] = v[a];
    }
}
hipError_t launch_debug_fft(const float2 *in, float2 *out, int inverse, int batch, const float2 *tw,
                            hipStream_t st)
{
    const size_t shmem = (kTabComplex + kWaveLdsComplex) * sizeof(float2);
    hipLaunchKernelGGL(k_debug_fft, dim3(batch), dim3(64), shmem, st, in, out, inverse, tw);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// small utilities
// ---------------------------------------------------------------------------------
__global__ void k_zero_tail_component(float2 *tails, long long n_streams, int which, int comp)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over n_streams * 512
    if (i >= n_streams * 512) return;
    float2 *p = tails + ((i >> 9) * 2 + which) * 512 + (i & 511);
    if (comp == 0) p->x = 0.0f; else p->y = 0.0f;
}
hipError_t launch_zero_tail_component(float2 *tails, long long n_streams, int which, int comp, hipStream_t st)
{
    const long long n = n_streams * 512;
    hipLaunchKernelGGL(k_zero_tail_component, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tails,
                       n_streams, which, comp);
    return hipGetLastError();
}

__global__ void k_scale_copy(const float *__restrict__ src, float *__restrict__ dst, long long n, float gain, int fp_mode)
{
    ohs_set_fp_mode(fp_mode);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] * gain;
}
hipError_t launch_scale_copy(const float *src, float *dst, long long n, float gain, hipStream_t st, int fp_mode)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_scale_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n, gain, fp_mode);
    return hipGetLastError();
}

// The same copy as the LAST kernel of a host call: the workgroup that finishes last stores `seq` into the call's
// completion word in pinned host memory (see k_engine_call).  `counter` is a device word that is zero between calls.
__global__ void k_scale_copy_done(const float *__restrict__ src, float *__restrict__ dst, long long n, float gain,
                                  unsigned *counter, unsigned *done, unsigned seq, int fp_mode)
{
    ohs_set_fp_mode(fp_mode);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] * gain;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");           // this workgroup's frames are out
        const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
hipError_t launch_scale_copy_done(const float *src, float *dst, long long n, float gain, unsigned *counter, unsigned *done,
                                  unsigned seq, hipStream_t st, int fp_mode)
{
    if (n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_scale_copy_done, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n, gain, counter,
                       done, seq, fp_mode);
    return hipGetLastError();
}

}  // namespace ohs
