// ohs_api.hip -- implementation of the C ABI declared in include/ohs_hip.h.
//
// Host-side orchestration only: device memory, FIFOs, per-path bookkeeping and
// kernel launches.  Every sample of audio is computed by the HIP kernels in
// conv_kernels.hip / eq_kernels.hip; there is no CPU compute path and every
// create call fails with OHS_ERR_NO_DEVICE when no GPU is usable.
#include "../../include/ohs_hip.h"
#include "kernels.h"
#include "wave_fft.hpp"
#include "host_internal.h"
#include "tuning.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace ohs;

namespace {

constexpr int BS = OHS_BLOCK_SIZE;
constexpr int NF = OHS_FFT_SIZE;

thread_local std::string g_err;
#ifdef OHS_EXPERIMENTS
// experiments build: the k-th chunk launch sequence of ohs_batch_process from now fails as if HIP had refused it
// (ohs_debug_inject_batch_failure); the product build has no such path
std::atomic<int> g_inject_batch_failure{0};
#endif

int fail(int status, const std::string &msg)
{
    g_err = msg;
    return status;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(OHS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

// ---- per-device context: the master twiddle table ------------------------------
struct DeviceCtx {
    int device = -1;
    float2 *d_tw = nullptr;
    float2 *d_tw2048 = nullptr; // k_conv_p1_os: w2048^bin in the paired layout
    int *d_pairs = nullptr;     // k_conv_tp_mac: leaders of the spectrum position pairs {b, mirror(b)}
    int n_pairs = 0;
    int num_cus = 256;
};
std::mutex g_ctx_mu;
std::vector<DeviceCtx *> g_ctx;

int get_ctx(int device, DeviceCtx **out)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(OHS_ERR_NO_DEVICE, std::string("no usable HIP device (") +
                                           (e != hipSuccess ? hipGetErrorString(e) : "count == 0") +
                                           "); this library has no CPU fallback");
    if (device < 0 || device >= count) return fail(OHS_ERR_INVALID_ARG, "device index out of range");
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (DeviceCtx *c : g_ctx)
        if (c->device == device) { *out = c; return OHS_OK; }
    HIP_TRY(hipSetDevice(device));
    DeviceCtx *c = new (std::nothrow) DeviceCtx();
    if (!c) return fail(OHS_ERR_ALLOC, "out of host memory");
    c->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            c->num_cus = cus;
    }
    // tw[e] = exp(-2 pi i e / 1024), computed in f64 and rounded to f32 (as rustfft does)
    std::vector<float2> tw(NF);
    for (int k = 0; k < NF; ++k) {
        const double ang = -2.0 * 3.14159265358979323846264338327950288 * (double)k / (double)NF;
        tw[k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    // every kernel receives the EXPANDED, lane-ordered tables (8.5 KiB) and copies them to LDS
    std::vector<float2> expanded(kTabComplex);
    expand_twiddle_tables(tw.data(), expanded.data());
    if (hipMalloc(&c->d_tw, kTabComplex * sizeof(float2)) != hipSuccess) {
        delete c;
        return fail(OHS_ERR_HIP, "hipMalloc(twiddles) failed");
    }
    if (hipMemcpy(c->d_tw, expanded.data(), kTabComplex * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(c->d_tw);
        delete c;
        return fail(OHS_ERR_HIP, "hipMemcpy(twiddles) failed");
    }
    {
        std::vector<float2> t2(NF);
        os_build_twiddles(t2.data());
        if (hipMalloc(&c->d_tw2048, NF * sizeof(float2)) != hipSuccess ||
            hipMemcpy(c->d_tw2048, t2.data(), NF * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
            hipFree(c->d_tw);
            delete c;
            return fail(OHS_ERR_HIP, "w2048 table upload failed");
        }
    }
    {
        int pairs[NF];
        c->n_pairs = conv_tp_build_pairs(pairs);
        if (hipMalloc(&c->d_pairs, NF * sizeof(int)) != hipSuccess ||
            hipMemcpy(c->d_pairs, pairs, c->n_pairs * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            hipFree(c->d_tw);
            delete c;
            return fail(OHS_ERR_HIP, "pair table upload failed");
        }
    }
    g_ctx.push_back(c);
    *out = c;
    return OHS_OK;
}

// the MAC reads up to Ppad - 1 blocks back (zero C/D beyond the real partition count)
int Ppad_for_ring(int Pmax) { return (Pmax + 15) / 16 * 16; }

int next_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// ---- convolution state shared by the engine (1 stream) and batch (S streams) ----
// ---- resident real-time kernels vs. device-wide waits ------------------------------------------------------
// hipFree, hipDeviceSynchronize and hipHostFree wait for EVERY kernel on the device -- a resident real-time kernel
// (k_engine_resident) of some other engine included, which only leaves when its owner stops it or after ~2 s without
// a call.  Library code that is about to make such a wait opens a DeviceWideSection: it asks every resident kernel of
// the device to leave (the `stop` word of its mailbox; the kernel is gone a microsecond later, and its owner restarts
// it on its next call exactly as after an idle exit) and keeps new ones from starting until the section ends (rt_start
// answers OHS_RT_NO_SLOT meanwhile: that call is served launch per call, like an engine beyond the device's cap).
// Only the mailbox word of a foreign engine is touched, under the registry's mutex, which an engine's destruction
// takes before its mailbox is freed.
constexpr int kMaxDevices = 64;
struct RtRegistry {
    std::mutex mu;
    std::vector<ohs_engine *> running;     // engines whose resident kernel has been launched and not yet reaped
    int hold = 0;                           // open DeviceWideSections
};
static RtRegistry g_rt_reg[kMaxDevices];
void rt_request_stop_locked(ohs_engine *e);      // (defined with the mailbox protocol below)
struct DeviceWideSection {
    int dev = 0;
    DeviceWideSection()
    {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        RtRegistry &r = g_rt_reg[dev % kMaxDevices];
        std::lock_guard<std::mutex> lk(r.mu);
        ++r.hold;
        for (ohs_engine *e : r.running) rt_request_stop_locked(e);
    }
    ~DeviceWideSection()
    {
        RtRegistry &r = g_rt_reg[dev % kMaxDevices];
        std::lock_guard<std::mutex> lk(r.mu);
        --r.hold;
    }
    DeviceWideSection(const DeviceWideSection &) = delete;
    DeviceWideSection &operator=(const DeviceWideSection &) = delete;
};

struct ConvState {
    size_t S = 1;
    float2 *d_H[4] = {nullptr, nullptr, nullptr, nullptr};
    int P[4] = {1, 1, 1, 1};
    long long since[4] = {0, 0, 0, 0};
    float2 *d_hist = nullptr;       // [S][cap][1024]
    int cap = 1;
    unsigned long long cnt = 0;
    float2 *d_tails = nullptr;      // [S][2][8][64]
    // P = 1 fast path
    float2 *d_tails_alt = nullptr;  // second state slab (k_conv_p1 writes its new tails here)
    float2 *d_cd = nullptr;         // [2][1024] combined spectra C, D
    bool cd_valid = false;
    float2 *d_chunk_tails = nullptr;
    size_t chunk_tails_cap = 0;     // in chunks per stream
    // time-parallel long-IR path
    float2 *d_cdm = nullptr;        // [Ppad][2][1024] combined spectra per partition
    int cdm_ppad = 0;
    bool cdm_valid = false;
    float2 *d_W = nullptr;          // [S][w_blocks][1024] MAC output scratch
    size_t w_blocks = 0;
    float2 *d_W1 = nullptr;         // [S][1024]
    int fp_mode = 0;                // denormal handling of every launch on this state (kernels.h: ohs_set_fp_mode)
    int xcd_lo = 0, xcd_n = 8;      // XCDs the P = 1 kernels may use (kernels.h: xcd_grid); set per call by the batch
    // Lazy state of the P = 1 fast path (batch objects; kernels.h: ConvP1Args::merged_out).  While tails_lazy, d_tails is
    // stale and the state is {d_merged, d_last_in}; conv_materialise_state turns it back into the per-path overlaps.
    bool lazy_ok = false, tails_lazy = false;
    float2 *d_merged = nullptr, *d_merged_alt = nullptr;    // [S][8][64]
    float *d_last_in = nullptr;                             // [S][2][512]
    // overlap-save plan of the P = 1 path (conv_os_kernels.hip; batch objects): the impulse responses in the time domain
    // (512 floats per path, zero-padded; valid while that path has one partition) and the (C, D) tables built from them
    float *d_irt = nullptr;         // [4][512]
    float4 *d_cd_os = nullptr;      // [2][1024]
    bool cd_os_valid = false;
    int conv_plan = 0;              // ohs_batch_set_conv_plan: 0 = the library's choice, 1 = block 512 / FFT 1024, 2 = hop 1536 / FFT 2048
    // ohs_*_set_speakers: what it last loaded into each path (a set_ir from anywhere else forgets it), so that a
    // change of the speaker angles re-loads only the paths whose impulse response really changed
    std::vector<float> spk_ir[4];
    bool spk_set[4] = {false, false, false, false};
};

void conv_free(ConvState &c)
{
    DeviceWideSection dws;      // (frees / device-wide waits below: no resident kernel may be in their way)
    for (int p = 0; p < 4; ++p) { if (c.d_H[p]) hipFree(c.d_H[p]); c.d_H[p] = nullptr; }
    if (c.d_hist) hipFree(c.d_hist);
    if (c.d_tails) hipFree(c.d_tails);
    if (c.d_tails_alt) hipFree(c.d_tails_alt);
    if (c.d_cd) hipFree(c.d_cd);
    if (c.d_chunk_tails) hipFree(c.d_chunk_tails);
    if (c.d_cdm) hipFree(c.d_cdm);
    if (c.d_W) hipFree(c.d_W);
    if (c.d_W1) hipFree(c.d_W1);
    if (c.d_merged) hipFree(c.d_merged);
    if (c.d_merged_alt) hipFree(c.d_merged_alt);
    if (c.d_last_in) hipFree(c.d_last_in);
    if (c.d_irt) hipFree(c.d_irt);
    if (c.d_cd_os) hipFree(c.d_cd_os);
    c.d_irt = nullptr; c.d_cd_os = nullptr; c.cd_os_valid = false;
    c.d_merged = nullptr; c.d_merged_alt = nullptr; c.d_last_in = nullptr; c.tails_lazy = false; c.lazy_ok = false;
    c.d_cdm = nullptr; c.d_W = nullptr; c.d_W1 = nullptr;
    c.d_hist = nullptr; c.d_tails = nullptr; c.d_tails_alt = nullptr; c.d_cd = nullptr; c.d_chunk_tails = nullptr;
}

int conv_init(ConvState &c, size_t S, hipStream_t st)
{
    c.S = S;
    for (int p = 0; p < 4; ++p) {
        // default path: one all-zero partition (convolution.rs:46-63); FFT(0) == 0
        HIP_TRY(hipMalloc(&c.d_H[p], NF * sizeof(float2)));
        HIP_TRY(hipMemsetAsync(c.d_H[p], 0, NF * sizeof(float2), st));
        c.P[p] = 1;
        c.since[p] = 0;
    }
    c.cap = 1;
    c.cnt = 0;
    HIP_TRY(hipMalloc(&c.d_hist, S * (size_t)c.cap * NF * sizeof(float2)));
    HIP_TRY(hipMemsetAsync(c.d_hist, 0, S * (size_t)c.cap * NF * sizeof(float2), st));
    HIP_TRY(hipMalloc(&c.d_tails, S * 2 * 8 * 64 * sizeof(float2)));
    HIP_TRY(hipMemsetAsync(c.d_tails, 0, S * 2 * 8 * 64 * sizeof(float2), st));
    HIP_TRY(hipMalloc(&c.d_tails_alt, S * 2 * 8 * 64 * sizeof(float2)));
    HIP_TRY(hipMalloc(&c.d_cd, 2 * NF * sizeof(float2)));
    c.cd_valid = false;
    return OHS_OK;
}

// Batch objects let the P = 1 fast path leave its state behind lazily: a k_conv_p1 launch then ends with the merged
// overlap of its last block and a copy of that block's input instead of a k_conv_p1_state launch (9 us + a launch gap
// per call, hidden nowhere when the convolution runs alone); the next fast launch starts from the merged overlap -- which
// also makes a call boundary invisible in the output bits.  (Tuning::p1_lazy_state = 0: the per-path state after every launch.)
int conv_enable_lazy_state(ConvState &c)
{
    if (!tuning().p1_lazy_state) return OHS_OK;
    HIP_TRY(hipMalloc(&c.d_merged, c.S * 8 * 64 * sizeof(float2)));
    HIP_TRY(hipMalloc(&c.d_merged_alt, c.S * 8 * 64 * sizeof(float2)));
    HIP_TRY(hipMalloc(&c.d_last_in, c.S * 2 * BS * sizeof(float)));
    HIP_TRY(hipMalloc(&c.d_irt, 4 * BS * sizeof(float)));
    HIP_TRY(hipMemset(c.d_irt, 0, 4 * BS * sizeof(float)));        // (default paths: one all-zero partition)
    HIP_TRY(hipMalloc(&c.d_cd_os, 2 * NF * sizeof(float4)));
    c.cd_os_valid = false;
    c.lazy_ok = true;
    c.tails_lazy = false;
    return OHS_OK;
}

// the per-path overlaps (d_tails) from the lazy state: whoever reads or edits d_tails calls this first -- a per-path
// set_ir (with the OLD spectra still in place: the overlaps belong to them), the general and the long-IR kernels
int conv_materialise_state(ConvState &c, DeviceCtx *ctx, hipStream_t st)
{
    if (!c.tails_lazy) return OHS_OK;
    ConvP1Args a;
    std::memset(&a, 0, sizeof(a));
    a.in = c.d_last_in; a.in_stream_stride = 2 * (long long)BS; a.in_ch_stride = (long long)BS;
    a.n_blocks = 1; a.n_streams = (int)c.S;
    for (int p = 0; p < 4; ++p) a.H[p] = c.d_H[p];
    a.tails_out = c.d_tails;
    a.tw = ctx->d_tw; a.fp_mode = c.fp_mode;
    a.xcd_lo = 0; a.xcd_n = 8;
    hipError_t e = launch_conv_p1_state(a, st);
    if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("conv_p1_state launch: ") + hipGetErrorString(e));
    c.tails_lazy = false;
    return OHS_OK;
}

// grow the spectrum ring so that cap >= need, keeping the most recent old_cap blocks
int conv_grow_ring(ConvState &c, int need, hipStream_t st)
{
    if (need <= c.cap) return OHS_OK;
    DeviceWideSection dws;      // (frees / device-wide waits below: no resident kernel may be in their way)
    const int ncap = next_pow2(need);
    float2 *nh = nullptr;
    HIP_TRY(hipMalloc(&nh, c.S * (size_t)ncap * NF * sizeof(float2)));
    HIP_TRY(hipMemsetAsync(nh, 0, c.S * (size_t)ncap * NF * sizeof(float2), st));
    for (int i = 0; i < c.cap; ++i) {
        if (c.cnt < (unsigned long long)(i + 1)) break;
        const unsigned long long u = c.cnt - 1 - (unsigned long long)i;
        const size_t so = (size_t)(u & (unsigned long long)(c.cap - 1));
        const size_t sn = (size_t)(u & (unsigned long long)(ncap - 1));
        // one strided copy over all streams
        HIP_TRY(hipMemcpy2DAsync(nh + sn * NF, (size_t)ncap * NF * sizeof(float2),
                                 c.d_hist + so * NF, (size_t)c.cap * NF * sizeof(float2),
                                 NF * sizeof(float2), c.S, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    hipFree(c.d_hist);
    c.d_hist = nh;
    c.cap = ncap;
    return OHS_OK;
}

// set_ir (convolution.rs:111-139) for one path; `ir` is a host pointer, or a device pointer on the state's
// device when ir_on_device (the node-level batch hands over what RCCL has just broadcast)
int conv_set_ir(ConvState &c, DeviceCtx *ctx, int path, const float *ir, size_t len, hipStream_t st,
                bool ir_on_device = false)
{
    if (path < 0 || path > 3) return fail(OHS_ERR_INVALID_ARG, "path must be 0..3");
    if (len && !ir) return fail(OHS_ERR_INVALID_ARG, "ir is NULL");
    if (len > (size_t)1 << 30) return fail(OHS_ERR_INVALID_ARG, "ir too long");
    const int P = len == 0 ? 1 : (int)((len + BS - 1) / BS);
    DeviceWideSection dws;      // (frees / device-wide waits below: no resident kernel may be in their way)
    {   // the overlaps at rest belong to the spectra that are about to be replaced
        const int rcm = conv_materialise_state(c, ctx, st);
        if (rcm) return rcm;
    }
    // every failure path below frees what it allocated: a HIP_TRY inside the lambda returns from the
    // lambda only, the buffers are released after it
    float2 *nH = nullptr;
    float *d_ir = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(&nH, (size_t)P * NF * sizeof(float2)));
        if (len == 0) {
            HIP_TRY(hipMemsetAsync(nH, 0, NF * sizeof(float2), st));   // :114-118 mute
        } else {
            const float *src = ir;
            if (!ir_on_device) {
                HIP_TRY(hipMalloc(&d_ir, len * sizeof(float)));
                HIP_TRY(hipMemcpyAsync(d_ir, ir, len * sizeof(float), hipMemcpyHostToDevice, st));
                src = d_ir;
            }
            HIP_TRY(launch_ir_spectrum(src, (int)len, P, nH, ctx->d_tw, st));
            if (c.d_irt && P == 1) {        // the overlap-save plan builds its 2048-point tables from the taps themselves
                HIP_TRY(hipMemsetAsync(c.d_irt + path * BS, 0, BS * sizeof(float), st));
                HIP_TRY(hipMemcpyAsync(c.d_irt + path * BS, src, len * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
        }
        if (c.d_irt && len == 0) HIP_TRY(hipMemsetAsync(c.d_irt + path * BS, 0, BS * sizeof(float), st));
        int rc = conv_grow_ring(c, P, st);
        if (rc) return rc;
        // one wait: the spectrum kernel has read d_ir, and nothing queued earlier on `st` still reads the
        // old d_H[path]
        HIP_TRY(hipStreamSynchronize(st));
        return OHS_OK;
    };
    const int rc0 = body();
    if (d_ir) hipFree(d_ir);
    if (rc0) { if (nH) hipFree(nH); return rc0; }
    hipFree(c.d_H[path]);
    c.d_H[path] = nH;
    c.P[path] = P;
    c.spk_set[path] = false;
    c.cd_os_valid = false;
    c.cd_valid = false;
    c.cdm_valid = false;
    c.since[path] = 0;      // :135-137 that path's history is "all zero" again
    if (c.since[0] == 0 && c.since[1] == 0 && c.since[2] == 0 && c.since[3] == 0) {
        // every path has forgotten its past: the ring holds nothing any path may still see.  Zeroing
        // it lets the gate-free time-parallel kernels run right away (all `since` equal).
        HIP_TRY(hipMemsetAsync(c.d_hist, 0, c.S * (size_t)c.cap * NF * sizeof(float2), st));
    }
    // :138 that path's overlap := 0.  Tails are (lsl + j lsr) and (rsl + j rsr).
    const int which = path >> 1, comp = path & 1;
    HIP_TRY(launch_zero_tail_component(c.d_tails, (long long)c.S, which, comp, st));
    HIP_TRY(hipStreamSynchronize(st));
    return OHS_OK;
}

// Chunks per stream of the P = 1 kernel: at least ONE round of resident waves (CUs x 16), and a wave count S * K that
// FILLS its last round -- all waves of a launch take the same time, so 2.13 rounds cost 3 (1024 and 2048 streams with
// K = ceil(2 R / S) ran at 71 % for that reason).  K is the smallest value from ceil(R / S) upwards whose last round is
// >= 97 % full.  (Rounds 1 and 2 asked for two rounds; since the chunk lengths follow the waves' age ranks one exact
// round is the faster plan -- config 3 without EQ: 4096 waves 0.537-0.541 ms, 8192 waves 0.549-0.555, 3840 0.563,
// 4352 0.81 -- and the pre-pass has half the boundary tails to compute.  Tuning::p1_rounds = 2 restores the old rule.)
long long conv_p1_chunks(const DeviceCtx *ctx, size_t S, long long n_blocks, long long target_override, int xcd_n = 8)
{
    const long long R = (long long)ctx->num_cus * xcd_n / 8 * conv_p1_waves_per_cu();
    long long K;
    if (target_override > 0) {
        K = (target_override + (long long)S - 1) / (long long)S;
    } else {
        const long long rounds = tuning().p1_rounds;
        const long long k0 = std::max<long long>(1, (rounds * R + (long long)S - 1) / (long long)S);
        K = k0;
        double best = 0.0;
        for (long long k = k0; k < k0 + 32; ++k) {
            const double rd = (double)((long long)S * k) / (double)R;
            const double eff = rd / std::ceil(rd - 1e-9);
            if (eff > best + 1e-9) { best = eff; K = k; }
            if (eff >= 0.97) break;
        }
    }
    // One chunk per stream (4096 streams and more) leaves the age ranks nothing to balance: two chunks, dealt out over
    // ranks {0, 3} / {1, 2} (conv_kernels.hip: p1_wave_job), cost a second round but end together -- 4096 streams: 8.9 ->
    // 8.0 ms of convolution per step.
    if (K == 1 && n_blocks >= 16 && target_override <= 0 && conv_p1_waves_per_cu() == 16) K = 2;
    if (n_blocks < 4 * K && target_override <= 0) {
        // A short launch (the last, 2 % time chunk of an overlapped batch step: 19 blocks), a few blocks per wave at
        // best: every chunk boundary costs the pre-pass a block's work, and a wave count just above a round starts a second,
        // almost empty one.  Cost in block times: rounds x blocks per wave for the main kernel + rounds for the boundary
        // tails (one more block per wave where the chunks' own waves compute them -- K = 2, 4, 8, 16 --, else a pre-pass
        // launch: its rounds + about a block time of launch); the smallest K of minimal cost (19 blocks x 256 streams:
        // K = 16, one round of at most 2 + 1 blocks, no pre-pass).
        const bool own_ok = conv_p1_waves_per_cu() == 16;
        long long best_k = std::min(K, n_blocks), best_cost = -1;
        for (long long k = 1; k <= std::min(K, n_blocks); ++k) {
            const long long rounds_main = ((long long)S * k + R - 1) / R, per_wave = (n_blocks + k - 1) / k;
            long long cost;
            if (k == 1) cost = rounds_main * per_wave;
            else if (own_ok && (k == 2 || k == 4 || k == 8 || k == 16)) cost = rounds_main * (per_wave + 1);
            else cost = rounds_main * per_wave + ((long long)S * (k - 1) + R - 1) / R + 1;
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_k = k; }
        }
        K = best_k;
    }
    if (K > n_blocks) K = n_blocks;
    if (K < 1) K = 1;
    return K;
}

// ohs_batch_set_conv_plan(0): which P = 1 plan the library picks by itself
bool conv_plan_auto_is_os(size_t S, long long n_blocks)
{
    (void)S; (void)n_blocks;
    return false;
}

int conv_launch(ConvState &c, DeviceCtx *ctx, const float *in, long long in_ss, long long in_cs,
                float *out, long long out_ss, long long out_cs, int n_blocks, float gain, hipStream_t st,
                bool allow_fast = true, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr)
{
    // (ev_start / ev_stop: recorded at the start / completion of the call's launches)
    struct EvScope {        // the paths that do not carry the events in a dispatch: recorded around their launches
        hipStream_t st; hipEvent_t stop; bool armed;
        ~EvScope() { if (armed && stop) hipEventRecord(stop, st); }
    } ev_scope{st, ev_stop, false};
    if (n_blocks <= 0) {                    // nothing to do (the pre-pass would index block -1)
        if (ev_start) HIP_TRY(hipEventRecord(ev_start, st));
        if (ev_stop) HIP_TRY(hipEventRecord(ev_stop, st));
        return OHS_OK;
    }
    const bool p1 = c.P[0] == 1 && c.P[1] == 1 && c.P[2] == 1 && c.P[3] == 1;
    if (p1 && allow_fast) {
        // every path has one partition: time-parallel single-FFT-pair kernel
        if (!c.cd_valid) {
            hipError_t e = launch_build_cd(c.d_H[0], c.d_H[1], c.d_H[2], c.d_H[3], c.d_cd, st);
            if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("build_cd: ") + hipGetErrorString(e));
            c.cd_valid = true;
        }
        const Tuning &tn = tuning();
        // Two plans for this case (ohs_batch_set_conv_plan): block 512 / FFT 1024 overlap-add (k_conv_p1: the plan whose
        // output bits tests/golden/p1_output_sha256.txt holds) and hop 1536 / FFT 2048 overlap-save (k_conv_p1_os: a third
        // less transform and product work per frame; batch objects only -- it hands the lazy state over).
        const bool os_ok = c.lazy_ok && c.d_irt && n_blocks >= 3 && c.xcd_n == 8 && tn.p1_xcd_n == 0 &&
                           ((in_ss | in_cs | out_ss | out_cs) & 1) == 0 &&
                           ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 7) == 0 &&
                           (unsigned long long)n_blocks * BS + 1536ull < (1ull << 31);
        const bool use_os = os_ok && (c.conv_plan == 2 || (c.conv_plan == 0 && conv_plan_auto_is_os(c.S, n_blocks)));
        if (use_os) {
            if (!c.cd_os_valid) {
                hipError_t e = launch_os_tables(c.d_irt, ctx->d_tw, ctx->d_tw2048, c.d_cd_os, st);
                if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("os tables: ") + hipGetErrorString(e));
                c.cd_os_valid = true;
            }
            // hop ranges per stream: a divisor of the workgroup's wave count (all of a stream's in one workgroup); rounds of resident waves
            // x hops per wave (+ 1: the tables and the first window), the smallest count of minimal cost
            const long long R = (long long)ctx->num_cus * conv_os_waves_per_cu();
            const long long H = ((long long)n_blocks * BS + 1535) / 1536;
            int K = 1;
            long long best = -1;
            for (int k = 1; k <= conv_os_waves_per_cu() && k <= H; ++k) {
                if (conv_os_waves_per_cu() % k) continue;       // (all of a stream's chunks in one workgroup)
                const long long cost = (((long long)c.S * k + R - 1) / R) * ((H + k - 1) / k + 1);
                if (best < 0 || cost < best) { best = cost; K = k; }
            }
            ConvOsArgs o;
            o.in = in; o.out = out;
            o.in_stream_stride = in_ss; o.in_ch_stride = in_cs; o.out_stream_stride = out_ss; o.out_ch_stride = out_cs;
            o.n_blocks = n_blocks; o.n_streams = (int)c.S; o.chunks = K;
            o.CD = c.d_cd_os; o.tw2048 = ctx->d_tw2048;
            o.tails = c.d_tails; o.merged_in = c.tails_lazy ? c.d_merged : nullptr;
            o.merged_out = c.d_merged_alt; o.last_in = c.d_last_in;
            o.tw = ctx->d_tw; o.gain = gain; o.fp_mode = c.fp_mode;
            hipError_t e = launch_conv_p1_os(o, st, ev_start, ev_stop);
            if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("conv_p1_os launch: ") + hipGetErrorString(e));
            std::swap(c.d_merged, c.d_merged_alt);
            c.tails_lazy = true;
            c.cnt += (unsigned long long)n_blocks;
            for (int p = 0; p < 4; ++p) c.since[p] += n_blocks;
            return OHS_OK;
        }
        const long long target_override = tn.p1_target_waves;
        int xcd_lo = c.xcd_lo, xcd_n = c.xcd_n;
        if (tn.p1_xcd_n > 0) { xcd_lo = tn.p1_xcd_lo; xcd_n = tn.p1_xcd_n; }
        const long long K = conv_p1_chunks(ctx, c.S, n_blocks, target_override, xcd_n);
        if ((size_t)K > c.chunk_tails_cap) {
            DeviceWideSection dws;
            HIP_TRY(hipStreamSynchronize(st));
            if (c.d_chunk_tails) hipFree(c.d_chunk_tails);
            c.d_chunk_tails = nullptr; c.chunk_tails_cap = 0;
            HIP_TRY(hipMalloc(&c.d_chunk_tails, c.S * (size_t)K * 512 * sizeof(float2)));
            c.chunk_tails_cap = (size_t)K;
        }
        ConvP1Args a;
        a.in = in; a.out = out;
        a.in_stream_stride = in_ss; a.in_ch_stride = in_cs;
        a.out_stream_stride = out_ss; a.out_ch_stride = out_cs;
        a.n_blocks = n_blocks; a.n_streams = (int)c.S; a.chunks = (int)K;
        a.CD = c.d_cd;
        for (int p = 0; p < 4; ++p) a.H[p] = c.d_H[p];
        a.tails = c.d_tails; a.tails_out = c.d_tails_alt; a.chunk_tails = c.d_chunk_tails;
        a.merged_in = c.tails_lazy ? c.d_merged : nullptr;
        a.merged_out = c.lazy_ok ? c.d_merged_alt : nullptr;
        a.last_in = c.lazy_ok ? c.d_last_in : nullptr;
        a.tw = ctx->d_tw; a.gain = gain; a.fp_mode = c.fp_mode;
        a.xcd_lo = xcd_lo; a.xcd_n = xcd_n;
        {   // boundary tails by the chunks' own waves where a stream's chunks share a workgroup (else: the pre-pass)
            const bool allowed = conv_p1_waves_per_cu() == 16 && (K == 2 || K == 4 || K == 8 || K == 16);
            a.own_tails = (allowed && tn.p1_own_tails) ? 1 : 0;
        }
        a.stagger = tn.p1_stagger;
        a.prio_mode = tn.p1_prio;
        // Chunk lengths by the age rank of the wave that runs a chunk (p1_chunk_begin), equal lengths when chunks are
        // short.  In the plans that do not spread a stream's chunks over the ranks, ALL of a stream's chunks can sit on
        // heavy ranks: its shortest chunk is then n * wmin / (K * wmax) blocks, and that one must keep >= 2 blocks (a
        // chunk of zero blocks would make the boundary-tail code index block b0 - 1 = -1).
        const int *kw = tn.p1_weights;
        const int wmin = std::min(std::min(kw[0], kw[1]), std::min(kw[2], kw[3]));
        const int wmax = std::max(std::max(kw[0], kw[1]), std::max(kw[2], kw[3]));
        const bool weighted = (long long)n_blocks * wmin >= 2ll * K * wmax;
        for (int g = 0; g < 4; ++g) a.weights[g] = weighted ? kw[g] : 1;
        hipError_t e = launch_conv_p1(a, st, ev_start, ev_stop);
        if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("conv_p1 launch: ") + hipGetErrorString(e));
        if (c.lazy_ok) {
            std::swap(c.d_merged, c.d_merged_alt);
            c.tails_lazy = true;
        } else {
            std::swap(c.d_tails, c.d_tails_alt);
        }
    } else {
        if (ev_start) HIP_TRY(hipEventRecord(ev_start, st));
        ev_scope.armed = true;
        {   // these kernels read and write the per-path overlaps
            const int rcm = conv_materialise_state(c, ctx, st);
            if (rcm) return rcm;
        }
        int Pmax = 1;
        for (int p = 0; p < 4; ++p) Pmax = std::max(Pmax, c.P[p]);
        const bool since_equal = c.since[0] == c.since[1] && c.since[1] == c.since[2] && c.since[2] == c.since[3];
        bool since_old = true;
        for (int p = 0; p < 4; ++p) since_old = since_old && c.since[p] >= c.P[p] - 1;
        // gate-free condition of the time-parallel kernels: no path may have blocks in the ring that
        // it must not see (either every path is older than its own IR -- path p reads P[p] - 1 blocks back --, or all
        // were reset together: conv_set_ir zeroed the ring then)
        const bool gate_free = since_old || since_equal;
        // Otherwise -- a per-path set_ir in mid-stream -- the same three kernels run on a ring whose old blocks they
        // cannot reach (the call starts Ppad - 1 zeroed slots further on) and k_conv_tp_old adds what each path may
        // still see of them to the first Pmax - 1 blocks.  The first segment must reach beyond every old block
        // (n_blocks >= Pmax): the state it leaves behind is computed gate-free.  Shorter calls take the general kernel
        // (one wave per stream, blocks in sequence: 290 x slower on BASELINE config 4, which is why long calls do not).
        const bool tp_gated = allow_fast && !gate_free && n_blocks >= 4 && n_blocks >= Pmax;
        const bool tp = allow_fast && n_blocks >= 4 && (gate_free || tp_gated);
        auto general = [&](const float *gin, float *gout, int nb, unsigned long long cnt, int since_add,
                           int skip_store, int zero_in, float2 *tails_out) -> int {
            ConvGeneralArgs a;
            a.in = gin; a.out = gout;
            a.in_stream_stride = in_ss; a.in_ch_stride = in_cs;
            a.out_stream_stride = out_ss; a.out_ch_stride = out_cs;
            a.n_blocks = nb;
            for (int p = 0; p < 4; ++p) { a.H[p] = c.d_H[p]; a.P[p] = c.P[p]; a.since[p] = c.since[p] + since_add; }
            a.hist = c.d_hist; a.cap = c.cap; a.cnt = cnt;
            a.tails = c.d_tails; a.tails_out = tails_out;
            a.skip_store = skip_store; a.zero_tails_in = zero_in;
            a.tw = ctx->d_tw; a.gain = gain; a.fp_mode = c.fp_mode;
            hipError_t e = launch_conv_general(a, (int)c.S, st);
            if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("conv launch: ") + hipGetErrorString(e));
            return OHS_OK;
        };
        if (!tp) {
            int rc = general(in, out, n_blocks, c.cnt, 0, 0, 0, c.d_tails);
            if (rc) return rc;
        } else {
            // segment size: keep the MAC scratch around 128 MiB
            long long SEG = 16384 / (long long)c.S;
            SEG = std::max<long long>(8, std::min<long long>(SEG, 1024));
            if (tp_gated) SEG = std::max<long long>(SEG, Pmax);
            const long long seg_max = std::min<long long>(SEG, n_blocks);
            const int Ppad = Ppad_for_ring(Pmax);
            // (gated: old blocks + the zeroed gap + the segment live in the ring together)
            int rc = conv_grow_ring(c, (int)(seg_max + (tp_gated ? 2 : 1) * (Ppad - 1) + (tp_gated ? Pmax : 0)), st);
            if (rc) return rc;
            const unsigned long long cnt_old = c.cnt;
            if (tp_gated) {
                // the zeroed gap: ring slots cnt .. cnt + Ppad - 2 of every stream (two strided memsets when it wraps)
                const unsigned long long gap = (unsigned long long)(Ppad - 1), cap = (unsigned long long)c.cap;
                const unsigned long long s0 = c.cnt & (cap - 1), n0 = std::min(gap, cap - s0);
                const size_t pitch = (size_t)c.cap * NF * sizeof(float2);
                HIP_TRY(hipMemset2DAsync(c.d_hist + s0 * NF, pitch, 0, (size_t)n0 * NF * sizeof(float2), c.S, st));
                if (n0 < gap)
                    HIP_TRY(hipMemset2DAsync(c.d_hist, pitch, 0, (size_t)(gap - n0) * NF * sizeof(float2), c.S, st));
                c.cnt += gap;
            }
            if (!c.d_cdm || c.cdm_ppad != Ppad) {       // (a set_ir that keeps Ppad re-uses the buffer: no wait, no malloc)
                DeviceWideSection dws;
                HIP_TRY(hipStreamSynchronize(st));
                if (c.d_cdm) hipFree(c.d_cdm);
                c.d_cdm = nullptr; c.cdm_valid = false;
                HIP_TRY(hipMalloc(&c.d_cdm, (size_t)Ppad * 2 * NF * sizeof(float2)));
                c.cdm_ppad = Ppad;
            }
            if (!c.cdm_valid) {
                hipError_t e = launch_build_cd_multi(c.d_H[0], c.d_H[1], c.d_H[2], c.d_H[3], c.P[0], c.P[1],
                                                     c.P[2], c.P[3], Ppad, c.d_cdm, st);
                if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("build_cd_multi: ") + hipGetErrorString(e));
                c.cdm_ppad = Ppad; c.cdm_valid = true;
            }
            if ((size_t)seg_max > c.w_blocks) {
                DeviceWideSection dws;
                HIP_TRY(hipStreamSynchronize(st));
                if (c.d_W) hipFree(c.d_W);
                c.d_W = nullptr; c.w_blocks = 0;
                HIP_TRY(hipMalloc(&c.d_W, c.S * (size_t)seg_max * NF * sizeof(float2)));
                c.w_blocks = (size_t)seg_max;
            }
            if (!c.d_W1) HIP_TRY(hipMalloc(&c.d_W1, c.S * (size_t)NF * sizeof(float2)));
            long long done = 0;
            while (done < n_blocks) {
                const int seg = (int)std::min<long long>(SEG, n_blocks - done);
                ConvTpArgs a;
                a.in = in + done * BS; a.out = out + done * BS;
                a.in_stream_stride = in_ss; a.in_ch_stride = in_cs;
                a.out_stream_stride = out_ss; a.out_ch_stride = out_cs;
                a.n_streams = (int)c.S; a.seg_blocks = seg; a.n_mac = seg; a.Ppad = Ppad;
                a.CD = c.d_cdm; a.hist = c.d_hist; a.cap = c.cap; a.cnt = c.cnt + (unsigned long long)done;
                a.W = c.d_W; a.W1 = c.d_W1; a.tails = c.d_tails; a.tails_out = c.d_tails_alt;
                a.pairs = ctx->d_pairs; a.n_pairs = ctx->n_pairs;
                a.tw = ctx->d_tw; a.gain = gain; a.fp_mode = c.fp_mode;
                hipError_t e = launch_conv_tp_forward(a, st);
                if (e == hipSuccess) e = launch_conv_tp_mac(a, st);
                if (e == hipSuccess && tp_gated && done == 0 && Pmax > 1) {
                    ConvTpOldArgs o;
                    o.n_streams = (int)c.S; o.n_old = Pmax - 1; o.Pmax = Pmax;
                    for (int p = 0; p < 4; ++p) { o.H[p] = c.d_H[p]; o.P[p] = c.P[p]; o.since[p] = c.since[p]; }
                    o.hist = c.d_hist; o.cap = c.cap; o.cnt_old = cnt_old;
                    o.W = c.d_W; o.n_mac = seg; o.fp_mode = c.fp_mode;
                    e = launch_conv_tp_old(o, st);
                }
                if (e == hipSuccess) e = launch_conv_tp_inverse(a, st);   // block 0 reads the OLD state; new state -> the other slab
                if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("conv_tp: ") + hipGetErrorString(e));
                std::swap(c.d_tails, c.d_tails_alt);
                done += seg;
            }
        }
    }
    c.cnt += (unsigned long long)n_blocks;
    for (int p = 0; p < 4; ++p) c.since[p] += n_blocks;
    return OHS_OK;
}

// ---- EQ table / state shared by ohs_eq (1 stream) and ohs_batch -------------------
struct EqState {
    size_t nb = 0;
    size_t chains = 2;
    std::vector<float> coeffs;   // nb * 5  {b0,b1,b2,a1,a2}
    std::vector<int> en;         // nb
    float *d_state = nullptr;    // [chains][kEqStateSlots][2], indexed by band
    bool exact_specials = false; // ohs_*_set_eq_exact_specials: conveyor kernel, exact for -0.0 / around non-finite input
    int fp_mode = 0;             // denormal handling (kernels.h: ohs_set_fp_mode)
    int xcd_lo = 0, xcd_n = 8;   // XCDs the ring kernel's launches may use (kernels.h: xcd_grid); set per call by the batch
};

using ohs_host::rbj;

int eq_init(EqState &e, size_t nb, size_t chains, float fs, hipStream_t st)
{
    if (nb > OHS_MAX_EQ_BANDS) return fail(OHS_ERR_INVALID_ARG, "num_bands > OHS_MAX_EQ_BANDS");
    e.nb = nb; e.chains = chains;
    e.coeffs.assign(nb * 5, 0.0f);
    e.en.assign(nb, 0);
    float c[5];
    int rc = rbj(OHS_FILTER_PEAK, fs, 20.0f, 0.707f, 0.0f, c);   // parametric_eq.rs:63-76
    if (rc) return rc;
    for (size_t b = 0; b < nb; ++b) std::memcpy(&e.coeffs[5 * b], c, sizeof(c));
    const size_t n = chains * (size_t)kEqStateSlots * 2;
    HIP_TRY(hipMalloc(&e.d_state, n * sizeof(float)));
    HIP_TRY(hipMemsetAsync(e.d_state, 0, n * sizeof(float), st));
    return OHS_OK;
}

int eq_reset(EqState &e, hipStream_t st)
{
    const size_t n = e.chains * (size_t)kEqStateSlots * 2;
    HIP_TRY(hipMemsetAsync(e.d_state, 0, n * sizeof(float), st));
    return OHS_OK;
}

// the enabled bands, in cascade order, as ONE pass table; returns their number (the caller checks <= 16 / <= 12)
int eq_single_pass_table(const EqState &e, EqPassTable &t)
{
    std::memset(&t, 0, sizeof(t));
    int nbp = 0;
    size_t last = 0;
    for (size_t b = 0; b < e.nb; ++b) {
        if (!e.en[b]) continue;
        if (nbp < 16) {
            t.slot[nbp] = (int)b;
            t.b0[nbp] = e.coeffs[5 * b + 0]; t.b1[nbp] = e.coeffs[5 * b + 1]; t.b2[nbp] = e.coeffs[5 * b + 2];
            t.a1[nbp] = e.coeffs[5 * b + 3]; t.a2[nbp] = e.coeffs[5 * b + 4];
            last = b;
        }
        ++nbp;
    }
    for (int j = std::min(nbp, 16); j < 16; ++j) t.slot[j] = (int)last;    // unused lanes shadow the last band's slot (never stored)
    return nbp;
}

// run the cascade over n frames of `chains` chains, in place allowed
// ev_start / ev_stop (optional): recorded at the start of the first and the completion of the last pass
int eq_launch(EqState &e, const float *in, float *out, long long ss, long long cs, long long n,
              hipStream_t st, bool *did_anything = nullptr, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr)
{
    bool did = false;
    const float *src = in;
    // enabled bands only, in cascade order, 16 per pass; a disabled band is the identity and keeps its
    // state (parametric_eq.rs:118-120), so it is simply not given a lane
    std::vector<size_t> on;
    for (size_t b = 0; b < e.nb; ++b)
        if (e.en[b]) on.push_back(b);
    for (size_t p0 = 0; p0 < on.size(); p0 += 16) {
        EqPassTable t;
        std::memset(&t, 0, sizeof(t));
        const int nbp = (int)std::min<size_t>(16, on.size() - p0);
        for (int j = 0; j < 16; ++j) {
            const size_t b = on[p0 + (size_t)std::min(j, nbp - 1)];      // unused lanes shadow the last band's slot (never stored)
            t.slot[j] = (int)b;
            if (j < nbp) {
                t.b0[j] = e.coeffs[5 * b + 0]; t.b1[j] = e.coeffs[5 * b + 1]; t.b2[j] = e.coeffs[5 * b + 2];
                t.a1[j] = e.coeffs[5 * b + 3]; t.a2[j] = e.coeffs[5 * b + 4];
            }
        }
        int xcd_lo = e.xcd_lo, xcd_n = e.xcd_n;
        if (tuning().eq_xcd_n > 0) { xcd_lo = tuning().eq_xcd_lo; xcd_n = tuning().eq_xcd_n; }
        const bool first = p0 == 0, last = p0 + 16 >= on.size();
        hipError_t err = launch_eq_pass(src, out, ss, cs, n, (int)e.chains, t, nbp, e.d_state, st, e.exact_specials, e.fp_mode,
                                        xcd_lo, xcd_n, first ? ev_start : nullptr, last ? ev_stop : nullptr);
        if (err != hipSuccess) return fail(OHS_ERR_HIP, std::string("eq launch: ") + hipGetErrorString(err));
        src = out;
        did = true;
    }
    if (!did) {     // no enabled band: the events mark this point of the stream
        if (ev_start) HIP_TRY(hipEventRecord(ev_start, st));
        if (ev_stop) HIP_TRY(hipEventRecord(ev_stop, st));
    }
    if (did_anything) *did_anything = did;
    return OHS_OK;
}

// ---- growable planar device FIFO [2][cap] ---------------------------------------------
struct DevFifo {
    float *d = nullptr;
    size_t cap = 0, len = 0;
};

int fifo_reserve(DevFifo &f, size_t need, hipStream_t st)
{
    if (need <= f.cap) return OHS_OK;
    DeviceWideSection dws;      // (frees / device-wide waits below: no resident kernel may be in their way)
    size_t ncap = f.cap ? f.cap : 4 * BS;
    while (ncap < need) ncap *= 2;
    float *nd = nullptr;
    HIP_TRY(hipMalloc(&nd, 2 * ncap * sizeof(float)));
    if (f.len) {
        HIP_TRY(hipMemcpyAsync(nd, f.d, f.len * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(nd + ncap, f.d + f.cap, f.len * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (f.d) hipFree(f.d);
    f.d = nd;
    f.cap = ncap;
    return OHS_OK;
}

// drop the first n frames.  The survivors are copied through `scratch` (>= 2*cap floats are not
// needed: survivors < cap) because source and destination ranges may overlap.
int fifo_drop_front(DevFifo &f, size_t n, DevFifo &scratch, hipStream_t st)
{
    const size_t rem = f.len - n;
    if (rem == 0) { f.len = 0; return OHS_OK; }
    if (rem <= n) {   // ranges [n, n+rem) and [0, rem) do not overlap
        HIP_TRY(hipMemcpyAsync(f.d, f.d + n, rem * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(f.d + f.cap, f.d + f.cap + n, rem * sizeof(float), hipMemcpyDeviceToDevice, st));
        f.len = rem;
        return OHS_OK;
    }
    int rc = fifo_reserve(scratch, f.cap, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(scratch.d, f.d + n, rem * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(scratch.d + scratch.cap, f.d + f.cap + n, rem * sizeof(float), hipMemcpyDeviceToDevice, st));
    std::swap(f.d, scratch.d);
    std::swap(f.cap, scratch.cap);
    f.len = rem;
    scratch.len = 0;
    return OHS_OK;
}

}  // namespace

// =======================================================================================
// handles
// =======================================================================================
struct ohs_engine {
    int device = 0;
    DeviceCtx *ctx = nullptr;
    hipStream_t st = nullptr;
    ConvState conv;
    DevFifo in, out, scratch;
    float *d_stage = nullptr;   // [2][stage_cap] pop staging (gain applied here)
    size_t stage_cap = 0;
    // single-launch call path (f2): pinned host staging mapped into the device + EQ output buffer
    float *h_in = nullptr, *h_out = nullptr;      // host [2][hcap]
    float *dm_in = nullptr, *dm_out = nullptr;    // the same memory, device addresses
    float *d_eqbuf = nullptr;                     // device [2][hcap]
    unsigned *h_done = nullptr, *dm_done = nullptr;   // completion word of the single-launch call (pinned; host / device address)
    unsigned call_seq = 0;
    size_t hcap = 0;
    uint64_t silent_frames = 0;                   // frames answered with the zero fill of convolution.rs:176-181
    // resident real-time kernel (ohs_engine_set_realtime): mailbox in pinned mapped memory, its own stream
    bool rt_enabled = false, rt_running = false, rt_slot = false;   // rt_slot: counted in the device's resident-kernel cap
    bool rt_wedged = false;         // a resident kernel did not leave within 5 s of its stop request and may still be alive:
                                    // it holds pointers to this engine's state, its mailbox and its stream (rt_stop)
    RtMailbox *mb = nullptr, *mb_dev = nullptr;
    hipStream_t rt_st = nullptr;
    unsigned rt_seq = 0;
    unsigned rt_eq_version = 0;     // version of the EQ block in the mailbox (bumped whenever its content changes)
};

struct ohs_eq {
    int device = 0;
    DeviceCtx *ctx = nullptr;
    hipStream_t st = nullptr;
    EqState eq;
    float fs = 48000.0f;
    float *d_buf = nullptr;     // [2][buf_cap]
    size_t buf_cap = 0;
    float *h_pin = nullptr;     // [2][pin_cap] pinned + mapped host staging for real-time sized calls
    float *dm_pin = nullptr;    // its device address
    float *d_pinbuf = nullptr;  // [2][pin_cap] device side
    unsigned *h_done = nullptr, *dm_done = nullptr, *d_counter = nullptr;   // completion word (pinned) + its workgroup counter
    unsigned call_seq = 0;
    size_t pin_cap = 0;
};

struct ohs_biquad {          // BiquadFilter (parametric_eq.rs:46-123): one DF2T section, one (mono) chain
    int device = 0;
    DeviceCtx *ctx = nullptr;
    hipStream_t st = nullptr;
    EqState eq;                 // one band, one chain
    float *d_buf = nullptr;
    size_t cap = 0;
};

struct ohs_batch {
    int device = 0;
    DeviceCtx *ctx = nullptr;
    hipStream_t st = nullptr;   // for set-up work only
    ConvState conv;
    EqState eq;
    int eq_enable = 0;
    float gain = 1.0f;
    // A HIP failure in the middle of a processing call leaves the per-stream state advanced for some time chunks only:
    // the handle is marked failed and every later processing call is refused until ohs_batch_reset (which zeroes the
    // state) -- never a continuation on half-advanced state.
    bool failed = false;
    std::string fail_msg;
    hipStream_t st2 = nullptr;  // convolution of time chunk i runs here while the EQ of chunk i+1 runs on the caller's stream
    bool overlap = true;
    int xcd_split = 0;          // x > 0: overlapped calls run the EQ on XCDs [0, x) and the convolution on [x, 8)
    // optional per-kernel timing: (begin, end, kind) event pairs; kind 0 = EQ, 1 = convolution
    bool profiling = false;
    struct Span { hipEvent_t a, b; int kind; };
    std::vector<Span> spans;
    uint64_t prof_calls = 0;
    std::vector<hipEvent_t> ev_pool;
    std::vector<hipEvent_t> ev_inflight;   // ordering events of the current / previous calls
    // ohs_batch_process_deferred: the convolutions of the previous call may still be running on st2
    std::vector<hipEvent_t> chunk_done;    // one persistent event per time chunk, re-recorded on st2 by every such call
    int chunk_done_n = 0;                  // chunks of the pending call
    bool join_pending = false;
    const float *pend_out = nullptr;       // geometry of the pending call: per-chunk waits only if the next one matches
    size_t pend_blocks = 0, pend_ss = 0, pend_cs = 0;
    // ohs_batch_process_host: three staging slots [S][2][slot_frames], copy-in / compute / copy-out streams
    static constexpr int kHostSlots = 3;
    float *d_slot[kHostSlots] = {nullptr, nullptr, nullptr};
    size_t slot_frames = 0;
    hipStream_t st_h2d = nullptr, st_comp = nullptr, st_d2h = nullptr;
    hipEvent_t ev_h2d[kHostSlots] = {nullptr, nullptr, nullptr}, ev_comp[kHostSlots] = {nullptr, nullptr, nullptr},
               ev_d2h[kHostSlots] = {nullptr, nullptr, nullptr};
};

namespace {

int engine_stage_reserve(ohs_engine *e, size_t n)
{
    if (n <= e->stage_cap) return OHS_OK;
    DeviceWideSection dws;      // (frees / device-wide waits below: no resident kernel may be in their way)
    size_t ncap = e->stage_cap ? e->stage_cap : 4 * BS;
    while (ncap < n) ncap *= 2;
    if (e->d_stage) hipFree(e->d_stage);
    e->d_stage = nullptr; e->stage_cap = 0;
    HIP_TRY(hipMalloc(&e->d_stage, 2 * ncap * sizeof(float)));
    e->stage_cap = ncap;
    return OHS_OK;
}

// Completion wait for the real-time sized calls: poll the stream (no sleep / wake-up on the audio
// thread); fall back to the blocking wait after 5 ms.
hipError_t sync_spin(hipStream_t st)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) return hipStreamSynchronize(st);
    }
}

int engine_pinned_reserve(ohs_engine *e, size_t n)
{
    if (n <= e->hcap) return OHS_OK;
    DeviceWideSection dws;      // (frees / device-wide waits below: no resident kernel may be in their way)
    size_t ncap = e->hcap ? e->hcap : 2048;
    while (ncap < n) ncap *= 2;
    if (e->h_in) hipHostFree(e->h_in);
    if (e->h_out) hipHostFree(e->h_out);
    if (e->d_eqbuf) hipFree(e->d_eqbuf);
    e->h_in = e->h_out = e->d_eqbuf = nullptr; e->hcap = 0;
    HIP_TRY(hipHostMalloc((void **)&e->h_in, 2 * ncap * sizeof(float), hipHostMallocMapped));
    HIP_TRY(hipHostMalloc((void **)&e->h_out, 2 * ncap * sizeof(float), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&e->dm_in, e->h_in, 0));
    HIP_TRY(hipHostGetDevicePointer((void **)&e->dm_out, e->h_out, 0));
    HIP_TRY(hipMalloc(&e->d_eqbuf, 2 * ncap * sizeof(float)));
    if (!e->h_done) {
        HIP_TRY(hipHostMalloc((void **)&e->h_done, 64, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&e->dm_done, e->h_done, 0));
        *e->h_done = 0;
    }
    e->hcap = ncap;
    return OHS_OK;
}

// the whole process_block call -- FIFO adapter, up to 8 internal blocks, pop
// with gain -- as ONE kernel launch reading / writing pinned host memory directly (plus one EQ
// launch for the lib.rs chain).  Requires the invariant in.len < 512, which both call paths keep.
int engine_process_single_launch(ohs_engine *e, const float *in_l, const float *in_r, float *out_l,
                                 float *out_r, size_t n, ohs_eq *eq, float gain)
{
    hipStream_t st = e->st;
    int rc = engine_pinned_reserve(e, n);
    if (rc) return rc;
    {
        int maxP = 1;
        for (int p = 0; p < 4; ++p) maxP = std::max(maxP, e->conv.P[p]);
        rc = conv_grow_ring(e->conv, maxP + 7, st);      // normally a no-op: ohs_engine_set_ir has done it
        if (rc) return rc;
    }
    std::memcpy(e->h_in, in_l, n * sizeof(float));
    std::memcpy(e->h_in + e->hcap, in_r, n * sizeof(float));
    const float *src_l = e->dm_in, *src_r = e->dm_in + e->hcap;
    if (eq) {   // lib.rs:1194: EQ on this host block before the engine sees it
        bool any = false;
        for (size_t i = 0; i < eq->eq.nb; ++i) any = any || eq->eq.en[i];
        if (any) {
            // to the device first: the EQ kernel's one-group prefetch cannot cover a PCIe read
            HIP_TRY(launch_scale_copy(e->dm_in, e->d_eqbuf, (long long)(e->hcap + n), 1.0f, st));
            rc = eq_launch(eq->eq, e->d_eqbuf, e->d_eqbuf, 0, (long long)e->hcap, (long long)n, st);
            if (rc) return rc;
            src_l = e->d_eqbuf; src_r = e->d_eqbuf + e->hcap;
        }
    }
    const size_t k = (e->in.len + n) / BS;
    const bool emit = e->out.len + k * BS >= n;
    rc = fifo_reserve(e->in, 2 * BS, st);
    if (rc) return rc;
    rc = fifo_reserve(e->out, e->out.len + k * BS + 1, st);
    if (rc) return rc;
    ConvState &c = e->conv;
    ConvGeneralArgs a;
    a.in = nullptr; a.out = nullptr;
    a.in_stream_stride = a.in_ch_stride = a.out_stream_stride = a.out_ch_stride = 0;
    a.n_blocks = (int)k;
    for (int p = 0; p < 4; ++p) { a.H[p] = c.d_H[p]; a.P[p] = c.P[p]; a.since[p] = c.since[p]; }
    a.hist = c.d_hist; a.cap = c.cap; a.cnt = c.cnt;
    a.tails = c.d_tails; a.tails_out = c.d_tails; a.skip_store = 0; a.zero_tails_in = 0;
    a.tw = e->ctx->d_tw; a.gain = 1.0f; a.fp_mode = e->conv.fp_mode;
    EngineCallArgs g;
    g.in_fifo = e->in.d; g.in_cap = (int)e->in.cap; g.in_len = (int)e->in.len;
    g.out_fifo = e->out.d; g.out_cap = (int)e->out.cap; g.out_len = (int)e->out.len;
    g.src_l = src_l; g.src_r = src_r;
    g.host_l = e->dm_out; g.host_r = e->dm_out + e->hcap;
    g.n = (int)n; g.emit = emit ? 1 : 0; g.gain_pop = gain;
    const unsigned seq = ++e->call_seq;
    hipError_t err = launch_engine_call(a, g, st, e->dm_done, seq);
    if (err != hipSuccess) return fail(OHS_ERR_HIP, std::string("engine call launch: ") + hipGetErrorString(err));
    c.cnt += k;
    for (int p = 0; p < 4; ++p) c.since[p] += (long long)k;
    e->in.len = e->in.len + n - k * BS;
    e->out.len = emit ? e->out.len + k * BS - n : e->out.len + k * BS;
    {   // the kernel's own completion word (it is stored behind a system-scope release of the frames); the stream
        // query is the fallback -- and the error path: a kernel that died never writes the word
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(e->h_done, __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
                HIP_TRY(hipStreamSynchronize(st));
                break;
            }
        }
    }
    if (emit) {
        std::memcpy(out_l, e->h_out, n * sizeof(float));
        std::memcpy(out_r, e->h_out + e->hcap, n * sizeof(float));
    } else {            // convolution.rs:176-181
        std::memset(out_l, 0, n * sizeof(float));
        std::memset(out_r, 0, n * sizeof(float));
        e->silent_frames += n;
    }
    return OHS_OK;
}

// ---- resident real-time kernel: host side of the mailbox protocol (k_engine_resident) --------------------
// A resident kernel holds one CU (149 KiB of LDS) for as long as it lives, so their number per device is capped:
// a quarter of the CUs by default (ohs_rt_configure).  Engines in real-time mode beyond the cap are served by
// the launch-per-call path until a slot frees up -- same results, never a wait for a CU that is not coming.
constexpr int OHS_RT_NO_SLOT = 1000;     // internal: rt_start found the device's cap reached (never crosses the C ABI)
static std::atomic<int> g_rt_resident[kMaxDevices];
static bool rt_slot_acquire(ohs_engine *e)
{
    if (e->rt_slot) return true;
    const int conf = tuning().rt_max_resident;
    const int cap = conf >= 0 ? conf : std::max(1, e->ctx->num_cus / 4);
    std::atomic<int> &c = g_rt_resident[e->device % kMaxDevices];
    if (c.fetch_add(1, std::memory_order_acq_rel) >= cap) { c.fetch_sub(1, std::memory_order_acq_rel); return false; }
    e->rt_slot = true;
    return true;
}
static void rt_mark_stopped(ohs_engine *e)
{
    e->rt_running = false;
    {
        RtRegistry &r = g_rt_reg[e->device % kMaxDevices];
        std::lock_guard<std::mutex> lk(r.mu);
        r.running.erase(std::remove(r.running.begin(), r.running.end(), e), r.running.end());
    }
    if (e->rt_slot) { g_rt_resident[e->device % kMaxDevices].fetch_sub(1, std::memory_order_acq_rel); e->rt_slot = false; }
}
static void rt_mark_stopped_unregistered(ohs_engine *e)     // registry mutex held by the caller, e not in the list
{
    e->rt_running = false;
    if (e->rt_slot) { g_rt_resident[e->device % kMaxDevices].fetch_sub(1, std::memory_order_acq_rel); e->rt_slot = false; }
}
void rt_request_stop_locked(ohs_engine *e)       // registry mutex held: e->mb is alive
{
    if (e->mb) __atomic_store_n(&e->mb->call_word[3], 1u, __ATOMIC_RELEASE);
}

void rt_stop(ohs_engine *e)
{
    if (!e->rt_running) return;
    __atomic_store_n(&e->mb->call_word[3], 1u, __ATOMIC_RELEASE);
    // the kernel sees `stop` at its next poll (~1 us) and every wave leaves.  Bounded: a kernel that does not leave
    // within 5 s is wedged, and an unbounded wait here would hang the audio thread with it.  A wedged kernel may
    // still be alive: it keeps its slot in the device's count and its place in the registry, the engine refuses
    // further calls (engine_process: the launch-per-call path would mutate state the kernel holds pointers to) and
    // its destruction leaks the mailbox and the stream instead of waiting on them -- until hipStreamQuery reports
    // that the kernel has gone after all (rt_wedged_recovered).
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(e->rt_st);
        if (q != hipErrorNotReady) break;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
            e->rt_enabled = false;
            e->rt_wedged = true;
            ohs_internal_set_error("resident kernel did not leave within 5 s of its stop request; the engine is unusable until it has");
            return;
        }
        std::this_thread::yield();
    }
    rt_mark_stopped(e);
}

// a wedged kernel that has left after all: the engine is usable again (launch per call; real-time mode stays off)
bool rt_wedged_recovered(ohs_engine *e)
{
    if (!e->rt_wedged) return true;
    if (hipStreamQuery(e->rt_st) == hipErrorNotReady) return false;
    e->rt_wedged = false;
    rt_mark_stopped(e);
    return true;
}

int rt_start(ohs_engine *e)
{
    if (e->rt_running) {
        if (__atomic_load_n(&e->mb->alive, __ATOMIC_ACQUIRE)) return OHS_OK;
        hipStreamSynchronize(e->rt_st);   // it left by its idle limit
        rt_mark_stopped(e);
    }
    if (!rt_slot_acquire(e)) return OHS_RT_NO_SLOT;
    if (!e->mb) {
        HIP_TRY(hipHostMalloc((void **)&e->mb, sizeof(RtMailbox), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&e->mb_dev, e->mb, 0));
        HIP_TRY(hipStreamCreateWithFlags(&e->rt_st, hipStreamNonBlocking));
        std::memset(e->mb, 0, sizeof(RtMailbox));
    }
    // everything queued on the engine's own stream (set_ir, a launch-per-call block) must have finished: the
    // resident kernel reads that state without any stream ordering
    HIP_TRY(hipStreamSynchronize(e->st));
    // the engine's state as it stands; the kernel advances the counters by itself from here (the host mirrors them)
    RtMailbox *mb = e->mb;
    ConvState &c = e->conv;
    ConvGeneralArgs &a = mb->conv;
    a.in = nullptr; a.out = nullptr;
    a.in_stream_stride = a.in_ch_stride = a.out_stream_stride = a.out_ch_stride = 0;
    a.n_blocks = 0;
    for (int p = 0; p < 4; ++p) { a.H[p] = c.d_H[p]; a.P[p] = c.P[p]; a.since[p] = c.since[p]; }
    a.hist = c.d_hist; a.cap = c.cap; a.cnt = c.cnt;
    a.tails = c.d_tails; a.tails_out = c.d_tails; a.skip_store = 0; a.zero_tails_in = 0;
    a.tw = e->ctx->d_tw; a.gain = 1.0f; a.fp_mode = e->conv.fp_mode;
    EngineCallArgs &g = mb->call;
    g.in_fifo = e->in.d; g.in_cap = (int)e->in.cap; g.in_len = (int)e->in.len;
    g.out_fifo = e->out.d; g.out_cap = (int)e->out.cap; g.out_len = (int)e->out.len;
    g.src_l = e->dm_in; g.src_r = e->dm_in + e->hcap;
    g.host_l = e->dm_out; g.host_r = e->dm_out + e->hcap;
    g.n = 0; g.emit = 0; g.gain_pop = 1.0f;
    mb->call_word[0] = e->rt_seq; mb->call_word[1] = 0; mb->call_word[2] = 0; mb->call_word[3] = 0;
    mb->done = e->rt_seq; mb->alive = 1;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    const unsigned long long idle_limit = tuning().rt_idle_polls;   // two reads in flight: one poll per ~0.6 us
    const unsigned sleep_arg = tuning().rt_poll_sleep;              // no pause between poll bursts: 0.3 us less per call
    {
        // launch and registration are one step under the registry's mutex: a DeviceWideSection either finds this
        // kernel in the list (and asks it to leave) or has raised `hold` before (and no kernel is started)
        RtRegistry &r = g_rt_reg[e->device % kMaxDevices];
        std::lock_guard<std::mutex> lk(r.mu);
        if (r.hold > 0) { rt_mark_stopped_unregistered(e); return OHS_RT_NO_SLOT; }
        hipError_t err = launch_engine_resident(e->mb_dev, e->ctx->d_tw, e->rt_seq, idle_limit, sleep_arg, e->rt_st);
        if (err != hipSuccess) {
            rt_mark_stopped_unregistered(e);
            return fail(OHS_ERR_HIP, std::string("resident kernel launch: ") + hipGetErrorString(err));
        }
        r.running.push_back(e);
    }
    e->rt_running = true;
    return OHS_OK;
}

// One process_block call through the mailbox.  Same FIFO arithmetic as engine_process_single_launch (the kernel
// runs it too); nothing is launched, allocated or freed here while the kernel is resident (growth stops it first).
int engine_process_resident(ohs_engine *e, const float *in_l, const float *in_r, float *out_l, float *out_r,
                            size_t n, float gain, EqState *eq, const EqPassTable *eq_tab, int eq_nb)
{
    ConvState &c = e->conv;
    const size_t k = (e->in.len + n) / BS;
    const bool emit = e->out.len + k * BS >= n;
    int maxP = 1;
    for (int p = 0; p < 4; ++p) maxP = std::max(maxP, c.P[p]);
    const bool grow = n > e->hcap || e->in.cap < (size_t)2 * BS || e->out.cap < e->out.len + k * BS + 1 || c.cap < maxP + 7;
    if (grow) {
        rt_stop(e);
        int rc = engine_pinned_reserve(e, std::max<size_t>(n, 8192));
        if (rc == OHS_OK) rc = fifo_reserve(e->in, 2 * BS, e->st);
        if (rc == OHS_OK) rc = fifo_reserve(e->out, std::max<size_t>(e->out.len + k * BS + 1, 8 * BS + 8192 + 1), e->st);
        if (rc == OHS_OK) rc = conv_grow_ring(c, maxP + 7, e->st);
        if (rc) return rc;
    }
    int rc = rt_start(e);
    if (rc) return rc;
    std::memcpy(e->h_in, in_l, n * sizeof(float));
    std::memcpy(e->h_in + e->hcap, in_r, n * sizeof(float));
    RtMailbox *mb = e->mb;
    if (eq) {   // the chain's EQ block: rewritten (and its version bumped) only when something in it changed
        if (mb->eq_nb != eq_nb || mb->eq_state != eq->d_state || mb->eq_buf != e->d_eqbuf ||
            mb->eq_buf_stride != (long long)e->hcap || std::memcmp(&mb->eq_tab, eq_tab, sizeof(EqPassTable)) != 0 ||
            e->rt_eq_version == 0) {
            mb->eq_nb = eq_nb; mb->eq_state = eq->d_state; mb->eq_buf = e->d_eqbuf; mb->eq_buf_stride = (long long)e->hcap;
            std::memcpy(&mb->eq_tab, eq_tab, sizeof(EqPassTable));
            mb->eq_version = ++e->rt_eq_version;
        }
    }
    const unsigned seq = ++e->rt_seq;
    unsigned gbits;
    std::memcpy(&gbits, &gain, sizeof(gbits));
    auto ring = [&]() {
        mb->call_word[1] = (unsigned)n | ((e->rt_eq_version & 0x7fffu) << 16) | (eq ? 0x80000000u : 0u);    // n <= 8192
        mb->call_word[2] = gbits;
        __atomic_store_n(&mb->call_word[0], seq, __ATOMIC_RELEASE);
    };
    ring();
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(&mb->done, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 1023u) == 0) {
            if (!__atomic_load_n(&mb->alive, __ATOMIC_ACQUIRE) && __atomic_load_n(&mb->done, __ATOMIC_ACQUIRE) != seq) {
                // the kernel left by its idle limit just as this call rang: start it again (from the state before
                // this call -- the host's counters have not been advanced yet), it then serves `seq`
                hipStreamSynchronize(e->rt_st);
                rt_mark_stopped(e);               // (off the registry, slot released: rt_start takes both again -- or
                                                  // answers OHS_RT_NO_SLOT / hold, and no stale entry stays behind)
                --e->rt_seq;                      // rt_start arms the mailbox with the last SERVED sequence number
                rc = rt_start(e);
                if (rc) return rc;                // (OHS_RT_NO_SLOT included: the caller serves this call launch per call)
                ++e->rt_seq;
                ring();
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                // (the host's counters have not been advanced: the call did not happen as far as the engine's
                // state on this side goes; rt_seq goes back to the last sequence number served)
                --e->rt_seq;
                e->rt_enabled = false;      // later calls: launch per call
                rt_stop(e);                 // bounded
                return fail(OHS_ERR_HIP, "resident kernel did not answer within 5 s; real-time mode switched off for this engine");
            }
        }
    }
    c.cnt += k;
    for (int p = 0; p < 4; ++p) c.since[p] += (long long)k;
    e->in.len = e->in.len + n - k * BS;
    e->out.len = emit ? e->out.len + k * BS - n : e->out.len + k * BS;
    if (emit) {
        std::memcpy(out_l, e->h_out, n * sizeof(float));
        std::memcpy(out_r, e->h_out + e->hcap, n * sizeof(float));
    } else {            // convolution.rs:176-181
        std::memset(out_l, 0, n * sizeof(float));
        std::memset(out_r, 0, n * sizeof(float));
        e->silent_frames += n;
    }
    return OHS_OK;
}

// process_block (convolution.rs:141-182) with the FIFOs on the device; optionally the
// lib.rs:1179-1207 chain around it (EQ on the freshly appended frames, gain on the popped ones).
int engine_process(ohs_engine *e, const float *in_l, const float *in_r, float *out_l, float *out_r,
                   size_t n, ohs_eq *eq, float gain)
{
    if (n == 0) return OHS_OK;
    if (!in_l || !in_r || !out_l || !out_r) return fail(OHS_ERR_INVALID_ARG, "NULL audio pointer");
    HIP_TRY(hipSetDevice(e->device));
    if (!rt_wedged_recovered(e))
        return fail(OHS_ERR_HIP, "a resident kernel of this engine did not leave and may still hold its state; the engine is unusable until it has");
    hipStream_t st = e->st;
    // real-time sized calls: the resident kernel's mailbox (real-time mode, no EQ in the chain) or everything in
    // one launch; long offline calls: the multi-kernel path below
    const bool rt_sized = (e->in.len + n) / BS <= 8 && n <= 8192 && e->in.len < (size_t)BS;
    // (the resident kernel gives every block of a call a pair of waves: up to kResidentMaxBlocks blocks)
    if (rt_sized && e->rt_enabled && (e->in.len + n) / BS <= (size_t)kResidentMaxBlocks) {
        // the resident kernel serves the call unless the chain's EQ needs more than the ring form's 12 bands, or the
        // device's cap of resident kernels is reached (OHS_RT_NO_SLOT: the launch-per-call path below serves it)
        int rrc = OHS_RT_NO_SLOT;
        bool tried = false;
        if (!eq) { rrc = engine_process_resident(e, in_l, in_r, out_l, out_r, n, gain, nullptr, nullptr, 0); tried = true; }
        else {
            EqPassTable t;
            const int nb_on = eq_single_pass_table(eq->eq, t);
            if (nb_on == 0) { rrc = engine_process_resident(e, in_l, in_r, out_l, out_r, n, gain, nullptr, nullptr, 0); tried = true; }
            else if (nb_on <= 12 && !eq->eq.exact_specials && eq->eq.fp_mode == e->conv.fp_mode) {     // (the resident kernel runs in the engine's mode)
                rrc = engine_process_resident(e, in_l, in_r, out_l, out_r, n, gain, &eq->eq, &t, nb_on); tried = true;
            }
        }
        if (tried && rrc != OHS_RT_NO_SLOT) return rrc;
    }
    rt_stop(e);         // the paths below launch kernels on the engine's state
    if (rt_sized) return engine_process_single_launch(e, in_l, in_r, out_l, out_r, n, eq, gain);
    int rc = fifo_reserve(e->in, e->in.len + n, st);
    if (rc) return rc;
    // :149-150 append
    float *app_l = e->in.d + e->in.len, *app_r = e->in.d + e->in.cap + e->in.len;
    HIP_TRY(hipMemcpyAsync(app_l, in_l, n * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(app_r, in_r, n * sizeof(float), hipMemcpyHostToDevice, st));
    if (eq) {   // lib.rs:1194: EQ in place on this host block, before the engine sees it
        rc = eq_launch(eq->eq, app_l, app_l, 0, (long long)e->in.cap, (long long)n, st);
        if (rc) return rc;
    }
    e->in.len += n;
    // :152-161 every complete 512-block
    const size_t k = e->in.len / BS;
    if (k) {
        rc = fifo_reserve(e->out, e->out.len + k * BS, st);
        if (rc) return rc;
        // a couple of blocks: one launch of the general kernel; long offline calls: time-parallel path
        rc = conv_launch(e->conv, e->ctx, e->in.d, 0, (long long)e->in.cap, e->out.d + e->out.len, 0,
                         (long long)e->out.cap, (int)k, 1.0f, st, /*allow_fast=*/k >= 4);
        if (rc) return rc;
        e->out.len += k * BS;
        rc = fifo_drop_front(e->in, k * BS, e->scratch, st);
        if (rc) return rc;
    }
    // :163-181 pop n or silence
    if (e->out.len >= n) {
        rc = engine_stage_reserve(e, n);
        if (rc) return rc;
        HIP_TRY(launch_scale_copy(e->out.d, e->d_stage, (long long)n, gain, st, e->conv.fp_mode));
        HIP_TRY(launch_scale_copy(e->out.d + e->out.cap, e->d_stage + e->stage_cap, (long long)n, gain, st, e->conv.fp_mode));
        HIP_TRY(hipMemcpyAsync(out_l, e->d_stage, n * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_r, e->d_stage + e->stage_cap, n * sizeof(float), hipMemcpyDeviceToHost, st));
        rc = fifo_drop_front(e->out, n, e->scratch, st);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(st));
    } else {
        HIP_TRY(hipStreamSynchronize(st));
        std::memset(out_l, 0, n * sizeof(float));
        std::memset(out_r, 0, n * sizeof(float));
        e->silent_frames += n;
    }
    return OHS_OK;
}

}  // namespace

// =======================================================================================
// C ABI
// =======================================================================================
extern "C" {

// shared with sofa_reader.cpp (not part of the public header)
void ohs_internal_set_error(const char *msg) { g_err = msg ? msg : ""; }

int ohs_version(void) { return (0 << 16) | (1 << 8) | 0; }

const char *ohs_status_string(int s)
{
    switch (s) {
    case OHS_OK: return "OHS_OK";
    case OHS_ERR_INVALID_ARG: return "OHS_ERR_INVALID_ARG";
    case OHS_ERR_NO_DEVICE: return "OHS_ERR_NO_DEVICE";
    case OHS_ERR_HIP: return "OHS_ERR_HIP";
    case OHS_ERR_OUTSIDE_NYQUIST: return "OHS_ERR_OUTSIDE_NYQUIST";
    case OHS_ERR_NEGATIVE_Q: return "OHS_ERR_NEGATIVE_Q";
    case OHS_ERR_ALLOC: return "OHS_ERR_ALLOC";
    case OHS_ERR_IO: return "OHS_ERR_IO";
    case OHS_ERR_PARSE: return "OHS_ERR_PARSE";
    case OHS_ERR_RCCL: return "OHS_ERR_RCCL";
    default: return "OHS_ERR_UNKNOWN";
    }
}

const char *ohs_last_error(void) { return g_err.c_str(); }

int ohs_device_count(int *count)
{
    if (!count) return fail(OHS_ERR_INVALID_ARG, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) {
        *count = 0;
        return fail(OHS_ERR_NO_DEVICE, std::string("no usable HIP device: ") +
                                           (e != hipSuccess ? hipGetErrorString(e) : "count == 0"));
    }
    *count = c;
    return OHS_OK;
}

// the two deployment options of the resident real-time kernels (include/ohs_hip.h)
int ohs_rt_configure(int max_resident_per_device, uint64_t idle_polls)
{
    Tuning &t = tuning_rt();
    t.rt_max_resident = max_resident_per_device;        // < 0: the default, a quarter of the device's CUs
    if (idle_polls) t.rt_idle_polls = idle_polls;
    return OHS_OK;
}

#ifdef OHS_EXPERIMENTS
// ---- experiments build only: not declared in include/ohs_hip.h, not exported by the product library ----
int ohs_debug_set_tuning(const char *key, const char *value)
{
    if (!key || !value) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    (void)tuning();        // (the environment first, so that it cannot overwrite this edit later)
    if (!tuning_set(key, value)) return fail(OHS_ERR_INVALID_ARG, std::string("unknown tuning key or bad value: ") + key + "=" + value);
    return OHS_OK;
}
int ohs_debug_inject_batch_failure(int kth_chunk)
{
    g_inject_batch_failure.store(kth_chunk);
    return OHS_OK;
}
#endif

int ohs_device_pci_bus_id(int device, char *out, size_t capacity)
{
    if (!out || capacity < 16) return fail(OHS_ERR_INVALID_ARG, "out is NULL or shorter than 16 bytes");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(OHS_ERR_NO_DEVICE, "no usable HIP device");
    if (device < 0 || device >= count) return fail(OHS_ERR_INVALID_ARG, "device index out of range");
    HIP_TRY(hipDeviceGetPCIBusId(out, (int)std::min<size_t>(capacity, 64), device));
    return OHS_OK;
}

// ---- diagnostics -----------------------------------------------------------------------
int ohs_debug_fft1024(int device, const float *in, float *out, int inverse, size_t batch)
{
    if (!in || !out || batch == 0 || batch > 65535) return fail(OHS_ERR_INVALID_ARG, "bad argument");
    DeviceCtx *ctx = nullptr;
    int rc = get_ctx(device, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    float2 *d = nullptr;
    const size_t bytes = batch * NF * sizeof(float2);
    HIP_TRY(hipMalloc(&d, 2 * bytes));
    hipError_t e = hipMemcpy(d, in, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_debug_fft(d, d + batch * NF, inverse, (int)batch, ctx->d_tw, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d + batch * NF, bytes, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail(OHS_ERR_HIP, hipGetErrorString(e));
    return OHS_OK;
}

// ---- engine ----------------------------------------------------------------------------
int ohs_engine_create(int device, ohs_engine **out)
{
    if (!out) return fail(OHS_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    DeviceCtx *ctx = nullptr;
    int rc = get_ctx(device, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    ohs_engine *e = new (std::nothrow) ohs_engine();
    if (!e) return fail(OHS_ERR_ALLOC, "out of host memory");
    e->device = device; e->ctx = ctx;
    if (hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking) != hipSuccess) {
        delete e;
        return fail(OHS_ERR_HIP, "hipStreamCreate failed");
    }
    rc = conv_init(e->conv, 1, e->st);
    if (rc == OHS_OK && hipStreamSynchronize(e->st) != hipSuccess) rc = fail(OHS_ERR_HIP, "sync failed");
    if (rc) { ohs_engine_destroy(e); return rc; }
    *out = e;
    return OHS_OK;
}

void ohs_engine_destroy(ohs_engine *e)
{
    if (!e) return;
    hipSetDevice(e->device);
    rt_stop(e);
    const bool wedged = !rt_wedged_recovered(e);
    if (wedged) {
        // the kernel may still poll its mailbox and write this engine's state: neither may be freed or waited for
        // (hipStreamDestroy / hipHostFree / hipFree wait for it).  Leak what it holds; the handle itself goes.
        {
            RtRegistry &r = g_rt_reg[e->device % kMaxDevices];
            std::lock_guard<std::mutex> lk(r.mu);
            r.running.erase(std::remove(r.running.begin(), r.running.end(), e), r.running.end());
        }
        ohs_internal_set_error("ohs_engine_destroy: a resident kernel never left; its mailbox, stream and device state were leaked");
        delete e;
        return;
    }
    rt_mark_stopped(e);     // (a slot taken by a start that then failed; also takes the engine off the registry)
    DeviceWideSection dws;  // (frees below: no resident kernel of another engine may be in their way)
    if (e->rt_st) hipStreamDestroy(e->rt_st);
    if (e->mb) hipHostFree(e->mb);
    if (e->st) hipStreamSynchronize(e->st);
    conv_free(e->conv);
    if (e->in.d) hipFree(e->in.d);
    if (e->out.d) hipFree(e->out.d);
    if (e->scratch.d) hipFree(e->scratch.d);
    if (e->d_stage) hipFree(e->d_stage);
    if (e->h_in) hipHostFree(e->h_in);
    if (e->h_out) hipHostFree(e->h_out);
    if (e->d_eqbuf) hipFree(e->d_eqbuf);
    if (e->h_done) hipHostFree(e->h_done);
    if (e->st) hipStreamDestroy(e->st);
    delete e;
}

int ohs_engine_clone(const ohs_engine *src, ohs_engine **out)
{
    if (!src || !out) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    ohs_engine *e = nullptr;
    int rc = ohs_engine_create(src->device, &e);
    if (rc) return rc;
    hipStream_t st = e->st;
    auto bail = [&](int code) { ohs_engine_destroy(e); return code; };
    const ConvState &s = src->conv;
    ConvState &d = e->conv;
    rt_stop(const_cast<ohs_engine *>(src));      // state at rest (the clone itself starts outside real-time mode)
    if (!rt_wedged_recovered(const_cast<ohs_engine *>(src)))
        return bail(fail(OHS_ERR_HIP, "a resident kernel of the source engine did not leave; its state cannot be copied"));
    DeviceWideSection dws;
    hipStreamSynchronize(src->st);
    e->silent_frames = src->silent_frames;
    d.fp_mode = s.fp_mode;
    for (int p = 0; p < 4; ++p) {
        hipFree(d.d_H[p]); d.d_H[p] = nullptr;
        if (hipMalloc(&d.d_H[p], (size_t)s.P[p] * NF * sizeof(float2)) != hipSuccess) return bail(fail(OHS_ERR_HIP, "hipMalloc"));
        if (hipMemcpyAsync(d.d_H[p], s.d_H[p], (size_t)s.P[p] * NF * sizeof(float2), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return bail(fail(OHS_ERR_HIP, "hipMemcpy"));
        d.P[p] = s.P[p]; d.since[p] = s.since[p];
    }
    hipFree(d.d_hist); d.d_hist = nullptr;
    d.cap = s.cap; d.cnt = s.cnt;
    if (hipMalloc(&d.d_hist, (size_t)s.cap * NF * sizeof(float2)) != hipSuccess) return bail(fail(OHS_ERR_HIP, "hipMalloc"));
    if (hipMemcpyAsync(d.d_hist, s.d_hist, (size_t)s.cap * NF * sizeof(float2), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return bail(fail(OHS_ERR_HIP, "hipMemcpy"));
    if (hipMemcpyAsync(d.d_tails, s.d_tails, 2 * 8 * 64 * sizeof(float2), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return bail(fail(OHS_ERR_HIP, "hipMemcpy"));
    const DevFifo *sf[2] = {&src->in, &src->out};
    DevFifo *df[2] = {&e->in, &e->out};
    for (int i = 0; i < 2; ++i) {
        if (!sf[i]->len) continue;
        rc = fifo_reserve(*df[i], sf[i]->len, st);
        if (rc) return bail(rc);
        hipMemcpyAsync(df[i]->d, sf[i]->d, sf[i]->len * sizeof(float), hipMemcpyDeviceToDevice, st);
        hipMemcpyAsync(df[i]->d + df[i]->cap, sf[i]->d + sf[i]->cap, sf[i]->len * sizeof(float), hipMemcpyDeviceToDevice, st);
        df[i]->len = sf[i]->len;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return bail(fail(OHS_ERR_HIP, "sync failed"));
    *out = e;
    return OHS_OK;
}

int ohs_engine_set_ir(ohs_engine *e, int path, const float *ir, size_t len)
{
    if (!e) return fail(OHS_ERR_INVALID_ARG, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    rt_stop(e);         // the resident kernel holds pointers to the spectra and the ring; the next call restarts it
    if (!rt_wedged_recovered(e)) return fail(OHS_ERR_HIP, "a resident kernel of this engine did not leave; the engine is unusable until it has");
    int rc = conv_set_ir(e->conv, e->ctx, path, ir, len, e->st);
    if (rc) return rc;
    // k_engine_call runs the (up to 8) blocks of a call side by side: the ring must hold the P - 1
    // older spectra the first of them reads AND the 8 new ones -- grown here, not on the audio thread
    int maxP = 1;
    for (int p = 0; p < 4; ++p) maxP = std::max(maxP, e->conv.P[p]);
    return conv_grow_ring(e->conv, maxP + 7, e->st);
}

int ohs_engine_process_block(ohs_engine *e, const float *in_l, const float *in_r, float *out_l,
                             float *out_r, size_t n)
{
    if (!e) return fail(OHS_ERR_INVALID_ARG, "engine is NULL");
    return engine_process(e, in_l, in_r, out_l, out_r, n, nullptr, 1.0f);
}

int ohs_engine_num_partitions(const ohs_engine *e, int path, size_t *out)
{
    if (!e || !out || path < 0 || path > 3) return fail(OHS_ERR_INVALID_ARG, "bad argument");
    *out = (size_t)e->conv.P[path];
    return OHS_OK;
}

int ohs_engine_set_flush_denormals(ohs_engine *e, int mode)
{
    if (!e) return fail(OHS_ERR_INVALID_ARG, "engine is NULL");
    if (mode < 0 || mode > 2) return fail(OHS_ERR_INVALID_ARG, "mode must be 0 (IEEE), 1 (FTZ) or 2 (FTZ | DAZ)");
    if (mode != e->conv.fp_mode) {
        HIP_TRY(hipSetDevice(e->device));
        rt_stop(e);         // the resident kernel set its mode when it started; the next call restarts it
        e->conv.fp_mode = mode;
    }
    return OHS_OK;
}

int ohs_engine_set_realtime(ohs_engine *e, int enable)
{
    if (!e) return fail(OHS_ERR_INVALID_ARG, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    e->rt_enabled = enable != 0;
    if (!e->rt_enabled) { rt_stop(e); return OHS_OK; }
    // everything a real-time sized call needs is reserved now, so that the audio thread neither allocates nor
    // launches: pinned staging for 8192 frames, FIFOs for eight blocks + a host block, the ring for P + 7 spectra
    int maxP = 1;
    for (int p = 0; p < 4; ++p) maxP = std::max(maxP, e->conv.P[p]);
    int rc = engine_pinned_reserve(e, 8192);
    if (rc == OHS_OK) rc = fifo_reserve(e->in, 2 * BS, e->st);
    if (rc == OHS_OK) rc = fifo_reserve(e->out, 8 * BS + 8192 + 1, e->st);
    if (rc == OHS_OK) rc = conv_grow_ring(e->conv, maxP + 7, e->st);
    if (rc) return rc;
    rc = rt_start(e);
    return rc == OHS_RT_NO_SLOT ? OHS_OK : rc;      // over the cap: the mode is on, calls go launch per call until a slot frees
}

// diagnostics: device timestamps (100 MHz) of the last call served by the resident kernel
int ohs_engine_debug_rt_stamps(const ohs_engine *e, uint64_t out[16])
{
    if (!e || !out || !e->mb) return fail(OHS_ERR_INVALID_ARG, "no resident kernel has run");
    for (int i = 0; i < 16; ++i) out[i] = e->mb->stamps[i];
    return OHS_OK;
}

int ohs_engine_silent_frames(const ohs_engine *e, uint64_t *frames)
{
    if (!e || !frames) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    *frames = e->silent_frames;
    return OHS_OK;
}

// The reference adds no latency when n is a multiple of 512; otherwise process_block answers with silence
// until its output FIFO holds n frames (convolution.rs:163-181), and every silent call delays the stream by n
// frames for good.  Pure FIFO arithmetic, simulated from an empty engine over several periods of the
// (n, 512) pattern.
int ohs_engine_latency_samples(const ohs_engine *e, size_t n, size_t *latency)
{
    (void)e;
    if (!latency) return fail(OHS_ERR_INVALID_ARG, "latency is NULL");
    if (n == 0 || n > ((size_t)1 << 30)) return fail(OHS_ERR_INVALID_ARG, "n out of range");
    size_t in_len = 0, out_len = 0, silent = 0;
    size_t g = n, h = BS;
    while (h) { const size_t t = g % h; g = h; h = t; }           // gcd(n, 512)
    const size_t period = (size_t)BS / g;                         // calls after which in_len repeats
    size_t since_silent = 0;
    for (size_t i = 0; i < 64 * period + 64 && since_silent <= 4 * period + 4; ++i) {
        in_len += n;
        const size_t k = in_len / BS;
        in_len -= k * BS;
        out_len += k * BS;
        if (out_len >= n) { out_len -= n; ++since_silent; }
        else { silent += n; since_silent = 0; }
    }
    *latency = silent;
    return OHS_OK;
}

// ---- eq --------------------------------------------------------------------------------
int ohs_biquad_coeffs(int filter_type, float fs, float fc, float q, float gain_db, float out[5])
{
    if (!out) return fail(OHS_ERR_INVALID_ARG, "out is NULL");
    return rbj(filter_type, fs, fc, q, gain_db, out);
}

int ohs_eq_create(int device, size_t num_bands, float fs, ohs_eq **out)
{
    if (!out) return fail(OHS_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    DeviceCtx *ctx = nullptr;
    int rc = get_ctx(device, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    ohs_eq *q = new (std::nothrow) ohs_eq();
    if (!q) return fail(OHS_ERR_ALLOC, "out of host memory");
    q->device = device; q->ctx = ctx; q->fs = fs;
    if (hipStreamCreateWithFlags(&q->st, hipStreamNonBlocking) != hipSuccess) {
        delete q;
        return fail(OHS_ERR_HIP, "hipStreamCreate failed");
    }
    rc = eq_init(q->eq, num_bands, 2, fs, q->st);
    if (rc == OHS_OK && hipStreamSynchronize(q->st) != hipSuccess) rc = fail(OHS_ERR_HIP, "sync failed");
    if (rc) { ohs_eq_destroy(q); return rc; }
    *out = q;
    return OHS_OK;
}

void ohs_eq_destroy(ohs_eq *q)
{
    if (!q) return;
    hipSetDevice(q->device);
    DeviceWideSection dws;
    if (q->st) hipStreamSynchronize(q->st);
    if (q->eq.d_state) hipFree(q->eq.d_state);
    if (q->d_buf) hipFree(q->d_buf);
    if (q->h_pin) hipHostFree(q->h_pin);
    if (q->d_pinbuf) hipFree(q->d_pinbuf);
    if (q->h_done) hipHostFree(q->h_done);
    if (q->d_counter) hipFree(q->d_counter);
    if (q->st) hipStreamDestroy(q->st);
    delete q;
}

int ohs_eq_update_band(ohs_eq *q, size_t band, float fs, int filter_type, float fc, float qv,
                       float gain_db, int enabled)
{
    if (!q) return fail(OHS_ERR_INVALID_ARG, "eq is NULL");
    if (band >= q->eq.nb) return OHS_OK;    // parametric_eq.rs:145 silently ignored
    float c[5];
    int rc = rbj(filter_type, fs, fc, qv, gain_db, c);
    if (rc) return rc;
    std::memcpy(&q->eq.coeffs[5 * band], c, sizeof(c));
    q->eq.en[band] = enabled != 0;
    return OHS_OK;
}

int ohs_eq_set_band_coeffs(ohs_eq *q, size_t band, const float coeffs[5], int enabled)
{
    if (!q || !coeffs) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    if (band >= q->eq.nb) return OHS_OK;
    std::memcpy(&q->eq.coeffs[5 * band], coeffs, 5 * sizeof(float));
    q->eq.en[band] = enabled != 0;
    return OHS_OK;
}

int ohs_eq_get_band_coeffs(const ohs_eq *q, size_t band, float coeffs[5], int *enabled)
{
    if (!q || !coeffs || band >= q->eq.nb) return fail(OHS_ERR_INVALID_ARG, "bad argument");
    std::memcpy(coeffs, &q->eq.coeffs[5 * band], 5 * sizeof(float));
    if (enabled) *enabled = q->eq.en[band];
    return OHS_OK;
}

int ohs_eq_process_block(ohs_eq *q, float *left, float *right, size_t n)
{
    if (!q) return fail(OHS_ERR_INVALID_ARG, "eq is NULL");
    if (n == 0) return OHS_OK;
    if (!left || !right) return fail(OHS_ERR_INVALID_ARG, "NULL audio pointer");
    HIP_TRY(hipSetDevice(q->device));
    if (n <= 8192) {
        // real-time sized call: copy kernels move the block between pinned, mapped host memory and the
        // device, the EQ runs on device memory -- three launches and one synchronisation, no copy engine.
        // (The EQ kernel does not touch host memory itself: its prefetch runs one 128-sample group = 1.5 us
        // ahead, less than a PCIe read, and its 64-byte stores are a poor fit for the bus.)
        // 1024 frames: 47 us per call (was 260 us with four hipMemcpyAsync of pageable memory).
        if (n > q->pin_cap) {
            DeviceWideSection dws;
            size_t ncap = q->pin_cap ? q->pin_cap : 2048;
            while (ncap < n) ncap *= 2;
            if (q->h_pin) hipHostFree(q->h_pin);
            if (q->d_pinbuf) hipFree(q->d_pinbuf);
            q->h_pin = q->dm_pin = q->d_pinbuf = nullptr; q->pin_cap = 0;
            HIP_TRY(hipHostMalloc((void **)&q->h_pin, 2 * ncap * sizeof(float), hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void **)&q->dm_pin, q->h_pin, 0));
            HIP_TRY(hipMalloc(&q->d_pinbuf, 2 * ncap * sizeof(float)));
            q->pin_cap = ncap;
        }
        std::memcpy(q->h_pin, left, n * sizeof(float));
        std::memcpy(q->h_pin + q->pin_cap, right, n * sizeof(float));
        bool any = false;
        for (size_t i = 0; i < q->eq.nb; ++i) any = any || q->eq.en[i];
        if (!any) return OHS_OK;                        // every band disabled: identity
        HIP_TRY(launch_scale_copy(q->dm_pin, q->d_pinbuf, (long long)(q->pin_cap + n), 1.0f, q->st));
        int rc = eq_launch(q->eq, q->d_pinbuf, q->d_pinbuf, 0, (long long)q->pin_cap, (long long)n, q->st);
        if (rc) return rc;
        if (!q->h_done) {
            HIP_TRY(hipHostMalloc((void **)&q->h_done, 64, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void **)&q->dm_done, q->h_done, 0));
            *q->h_done = 0;
            HIP_TRY(hipMalloc((void **)&q->d_counter, sizeof(unsigned)));
            HIP_TRY(hipMemsetAsync(q->d_counter, 0, sizeof(unsigned), q->st));
        }
        const unsigned seq = ++q->call_seq;
        HIP_TRY(launch_scale_copy_done(q->d_pinbuf, q->dm_pin, (long long)(q->pin_cap + n), 1.0f, q->d_counter, q->dm_done, seq, q->st));
        {   // the copy-out kernel's completion word; the stream is the fallback and the error path
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (__atomic_load_n(q->h_done, __ATOMIC_ACQUIRE) != seq) {
                if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
                    HIP_TRY(hipStreamSynchronize(q->st));
                    break;
                }
            }
        }
        std::memcpy(left, q->h_pin, n * sizeof(float));
        std::memcpy(right, q->h_pin + q->pin_cap, n * sizeof(float));
        return OHS_OK;
    }
    if (n > q->buf_cap) {
        DeviceWideSection dws;
        size_t ncap = q->buf_cap ? q->buf_cap : 4 * BS;
        while (ncap < n) ncap *= 2;
        if (q->d_buf) hipFree(q->d_buf);
        q->d_buf = nullptr; q->buf_cap = 0;
        HIP_TRY(hipMalloc(&q->d_buf, 2 * ncap * sizeof(float)));
        q->buf_cap = ncap;
    }
    HIP_TRY(hipMemcpyAsync(q->d_buf, left, n * sizeof(float), hipMemcpyHostToDevice, q->st));
    HIP_TRY(hipMemcpyAsync(q->d_buf + q->buf_cap, right, n * sizeof(float), hipMemcpyHostToDevice, q->st));
    int rc = eq_launch(q->eq, q->d_buf, q->d_buf, 0, (long long)q->buf_cap, (long long)n, q->st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(left, q->d_buf, n * sizeof(float), hipMemcpyDeviceToHost, q->st));
    HIP_TRY(hipMemcpyAsync(right, q->d_buf + q->buf_cap, n * sizeof(float), hipMemcpyDeviceToHost, q->st));
    HIP_TRY(hipStreamSynchronize(q->st));
    return OHS_OK;
}

int ohs_eq_set_flush_denormals(ohs_eq *q, int mode)
{
    if (!q) return fail(OHS_ERR_INVALID_ARG, "eq is NULL");
    if (mode < 0 || mode > 2) return fail(OHS_ERR_INVALID_ARG, "mode must be 0 (IEEE), 1 (FTZ) or 2 (FTZ | DAZ)");
    q->eq.fp_mode = mode;
    return OHS_OK;
}

int ohs_eq_set_exact_specials(ohs_eq *q, int enable)
{
    if (!q) return fail(OHS_ERR_INVALID_ARG, "eq is NULL");
    q->eq.exact_specials = enable != 0;
    return OHS_OK;
}

int ohs_eq_reset(ohs_eq *q)
{
    if (!q) return fail(OHS_ERR_INVALID_ARG, "eq is NULL");
    HIP_TRY(hipSetDevice(q->device));
    int rc = eq_reset(q->eq, q->st);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(q->st));
    return OHS_OK;
}

int ohs_eq_frequency_response(const ohs_eq *q, float fs, const float *freqs, size_t n, float *out)
{
    if (!q || (n && (!freqs || !out))) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    if (n == 0) return OHS_OK;
    HIP_TRY(hipSetDevice(q->device));
    const size_t nb = q->eq.nb;
    float *d = nullptr;
    const size_t words = nb * 5 + nb + 2 * n + 8;
    HIP_TRY(hipMalloc(&d, words * sizeof(float)));
    float *d_c = d, *d_f = d + nb * 5 + nb + 4, *d_o = d_f + n;
    int *d_en = (int *)(d + nb * 5);
    hipError_t e = hipSuccess;
    if (nb) {
        e = hipMemcpyAsync(d_c, q->eq.coeffs.data(), nb * 5 * sizeof(float), hipMemcpyHostToDevice, q->st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_en, q->eq.en.data(), nb * sizeof(int), hipMemcpyHostToDevice, q->st);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d_f, freqs, n * sizeof(float), hipMemcpyHostToDevice, q->st);
    if (e == hipSuccess) e = launch_eq_freq_response(d_c, d_en, (int)nb, fs, d_f, (int)n, d_o, q->st);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_o, n * sizeof(float), hipMemcpyDeviceToHost, q->st);
    if (e == hipSuccess) e = hipStreamSynchronize(q->st);
    { DeviceWideSection dws; hipFree(d); }
    if (e != hipSuccess) return fail(OHS_ERR_HIP, hipGetErrorString(e));
    return OHS_OK;
}

// ---- BiquadFilter as a type of its own (parametric_eq.rs:46-123) -------------------------------
int ohs_biquad_create(int device, float initial_sample_rate, ohs_biquad **out)
{
    if (!out) return fail(OHS_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    DeviceCtx *ctx = nullptr;
    int rc = get_ctx(device, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    ohs_biquad *f = new (std::nothrow) ohs_biquad();
    if (!f) return fail(OHS_ERR_ALLOC, "out of host memory");
    f->device = device; f->ctx = ctx;
    if (hipStreamCreateWithFlags(&f->st, hipStreamNonBlocking) != hipSuccess) {
        delete f;
        return fail(OHS_ERR_HIP, "hipStreamCreate failed");
    }
    rc = eq_init(f->eq, 1, 1, initial_sample_rate, f->st);        // PeakingEQ 0 dB @ 20 Hz Q 0.707, disabled (:63-76)
    if (rc == OHS_OK && hipStreamSynchronize(f->st) != hipSuccess) rc = fail(OHS_ERR_HIP, "sync failed");
    if (rc) { ohs_biquad_destroy(f); return rc; }
    *out = f;
    return OHS_OK;
}

void ohs_biquad_destroy(ohs_biquad *f)
{
    if (!f) return;
    hipSetDevice(f->device);
    DeviceWideSection dws;
    if (f->st) hipStreamSynchronize(f->st);
    if (f->eq.d_state) hipFree(f->eq.d_state);
    if (f->d_buf) hipFree(f->d_buf);
    if (f->st) hipStreamDestroy(f->st);
    delete f;
}

int ohs_biquad_clone(const ohs_biquad *src, ohs_biquad **out)       // impl Clone :52-60: coefficients, state, enabled
{
    if (!src || !out) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    ohs_biquad *f = nullptr;
    int rc = ohs_biquad_create(src->device, 48000.0f, &f);
    if (rc) return rc;
    f->eq.coeffs = src->eq.coeffs; f->eq.en = src->eq.en;
    f->eq.exact_specials = src->eq.exact_specials; f->eq.fp_mode = src->eq.fp_mode;
    hipStreamSynchronize(src->st);
    if (hipMemcpyAsync(f->eq.d_state, src->eq.d_state, (size_t)kEqStateSlots * 2 * sizeof(float), hipMemcpyDeviceToDevice,
                       f->st) != hipSuccess || hipStreamSynchronize(f->st) != hipSuccess) {
        ohs_biquad_destroy(f);
        return fail(OHS_ERR_HIP, "state copy failed");
    }
    *out = f;
    return OHS_OK;
}

int ohs_biquad_reset_state(ohs_biquad *f)
{
    if (!f) return fail(OHS_ERR_INVALID_ARG, "filter is NULL");
    HIP_TRY(hipSetDevice(f->device));
    int rc = eq_reset(f->eq, f->st);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(f->st));
    return OHS_OK;
}

int ohs_biquad_set_enabled(ohs_biquad *f, int enabled)
{
    if (!f) return fail(OHS_ERR_INVALID_ARG, "filter is NULL");
    f->eq.en[0] = enabled != 0;
    return OHS_OK;
}

int ohs_biquad_update_coeffs(ohs_biquad *f, int filter_type, float sample_rate, float center_freq, float q, float gain_db)
{
    if (!f) return fail(OHS_ERR_INVALID_ARG, "filter is NULL");
    float c[5];
    int rc = rbj(filter_type, sample_rate, center_freq, q, gain_db, c);
    if (rc) return rc;
    std::memcpy(&f->eq.coeffs[0], c, sizeof(c));        // the state is kept (update_coefficients, :112)
    return OHS_OK;
}

int ohs_biquad_set_coeffs(ohs_biquad *f, const float coeffs[5])
{
    if (!f || !coeffs) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    std::memcpy(&f->eq.coeffs[0], coeffs, 5 * sizeof(float));
    return OHS_OK;
}

int ohs_biquad_process(ohs_biquad *f, float *samples, size_t n)
{
    if (!f) return fail(OHS_ERR_INVALID_ARG, "filter is NULL");
    if (n == 0) return OHS_OK;
    if (!samples) return fail(OHS_ERR_INVALID_ARG, "samples is NULL");
    if (!f->eq.en[0]) return OHS_OK;                    // disabled: the input sample, bit for bit (:118-120)
    HIP_TRY(hipSetDevice(f->device));
    if (n > f->cap) {
        DeviceWideSection dws;
        size_t ncap = f->cap ? f->cap : 4 * BS;
        while (ncap < n) ncap *= 2;
        if (f->d_buf) hipFree(f->d_buf);
        f->d_buf = nullptr; f->cap = 0;
        HIP_TRY(hipMalloc(&f->d_buf, ncap * sizeof(float)));
        f->cap = ncap;
    }
    HIP_TRY(hipMemcpyAsync(f->d_buf, samples, n * sizeof(float), hipMemcpyHostToDevice, f->st));
    int rc = eq_launch(f->eq, f->d_buf, f->d_buf, 0, 0, (long long)n, f->st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(samples, f->d_buf, n * sizeof(float), hipMemcpyDeviceToHost, f->st));
    HIP_TRY(hipStreamSynchronize(f->st));
    return OHS_OK;
}

// ---- speaker angles -> four set_ir (the wiring the reference leaves open; speakers.cpp) ---------
}  // extern "C"
namespace {
template <class SetIr>
int set_speakers_impl(ConvState &c, const ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r, float radius_m,
                      float fs, unsigned *changed_mask, SetIr set_ir)
{
    if (changed_mask) *changed_mask = 0;
    if (!sofa) return fail(OHS_ERR_INVALID_ARG, "sofa is NULL");
    std::vector<float> irs[4];
    int rc = ohs_host::speaker_irs(sofa, az_l, el_l, az_r, el_r, radius_m, fs, irs);
    if (rc) return rc;
    for (int p = 0; p < 4; ++p) {
        if (c.spk_set[p] && c.spk_ir[p] == irs[p]) continue;        // same nearest measurement as before: nothing to do
        rc = set_ir(p, irs[p].data(), irs[p].size());
        if (rc) return rc;
        c.spk_ir[p].swap(irs[p]);
        c.spk_set[p] = true;
        if (changed_mask) *changed_mask |= 1u << p;
    }
    return OHS_OK;
}
}  // namespace
extern "C" {

int ohs_engine_set_speakers(ohs_engine *e, const ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r,
                            float radius_m, float fs, unsigned *changed_mask)
{
    if (!e) return fail(OHS_ERR_INVALID_ARG, "engine is NULL");
    return set_speakers_impl(e->conv, sofa, az_l, el_l, az_r, el_r, radius_m, fs, changed_mask,
                             [&](int p, const float *ir, size_t n) { return ohs_engine_set_ir(e, p, ir, n); });
}

int ohs_batch_set_speakers(ohs_batch *b, const ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r,
                           float radius_m, float fs, unsigned *changed_mask)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    return set_speakers_impl(b->conv, sofa, az_l, el_l, az_r, el_r, radius_m, fs, changed_mask,
                             [&](int p, const float *ir, size_t n) { return ohs_batch_set_ir(b, p, ir, n); });
}

// ---- chain -----------------------------------------------------------------------------
int ohs_chain_process(ohs_engine *e, ohs_eq *eq, int bypass, int eq_enable, float gain, float *left,
                      float *right, size_t n)
{
    if (!e) return fail(OHS_ERR_INVALID_ARG, "engine is NULL");
    if (bypass) return OHS_OK;                               // lib.rs:1169
    if (eq_enable && !eq) return fail(OHS_ERR_INVALID_ARG, "eq_enable set but eq is NULL");
    if (eq_enable && eq->device != e->device) return fail(OHS_ERR_INVALID_ARG, "eq and engine on different devices");
    return engine_process(e, left, right, left, right, n, eq_enable ? eq : nullptr, gain);
}

// ---- batch -----------------------------------------------------------------------------
int ohs_batch_create(int device, size_t n_streams, size_t num_bands, ohs_batch **out)
{
    if (!out) return fail(OHS_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n_streams == 0 || n_streams > (1u << 20)) return fail(OHS_ERR_INVALID_ARG, "n_streams out of range");
    DeviceCtx *ctx = nullptr;
    int rc = get_ctx(device, &ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    ohs_batch *b = new (std::nothrow) ohs_batch();
    if (!b) return fail(OHS_ERR_ALLOC, "out of host memory");
    b->device = device; b->ctx = ctx;
    if (hipStreamCreateWithFlags(&b->st, hipStreamNonBlocking) != hipSuccess) {
        delete b;
        return fail(OHS_ERR_HIP, "hipStreamCreate failed");
    }
    {
        hipError_t se = hipErrorUnknown;
        const Tuning &tn = tuning();
        if (!tn.conv_cu_mask.empty()) {      // (experiments: the overlapped convolution's stream confined to a CU set)
            se = hipExtStreamCreateWithCUMask(&b->st2, (uint32_t)tn.conv_cu_mask.size(), tn.conv_cu_mask.data());
            if (se != hipSuccess) fprintf(stderr, "[ohs] conv_cu_mask ignored: %s\n", hipGetErrorString(se));
        }
        if (se != hipSuccess) se = hipStreamCreateWithFlags(&b->st2, hipStreamNonBlocking);
        if (se != hipSuccess) {
            hipStreamDestroy(b->st);
            delete b;
            return fail(OHS_ERR_HIP, "hipStreamCreate failed");
        }
        // EQ || convolution overlap policy.  An EQ wave saturates the vector unit of its SIMD (every instruction of
        // the ring form is a 4-cycle DPP / packed operation: two EQ waves on one SIMD take 1.82x the time of one,
        // four 3.47x -- profiles/r03_eq_share.txt), so the convolution only makes progress on CUs that host no EQ
        // wave; from one EQ wave per SIMD on (almost) every CU upwards the two kernels merely get in each other's
        // way (2048 streams: 13.7-16.1 ms overlapped, 11.5 ms one after the other).  Overlap while at least an
        // eighth of the CUs stays free of EQ waves, serialise beyond.
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        const size_t eq_waves = (2 * n_streams + 3) / 4;            // 4 chains per wave
        const size_t eq_cus = (eq_waves + 3) / 4;                   // one wave per SIMD once there is a wave per CU
        b->overlap = eq_cus * 8 <= (size_t)cus * 7;
        if (tn.no_overlap) b->overlap = false;
        if (tn.force_overlap) b->overlap = true;
        b->xcd_split = tn.xcd_split;        // (experiments: EQ launches on XCDs [0, x), overlapped convolution on [x, 8))
    }
    rc = conv_init(b->conv, n_streams, b->st);
    if (rc == OHS_OK) rc = conv_enable_lazy_state(b->conv);
    if (rc == OHS_OK) rc = eq_init(b->eq, num_bands, 2 * n_streams, 48000.0f, b->st);
    if (rc == OHS_OK && hipStreamSynchronize(b->st) != hipSuccess) rc = fail(OHS_ERR_HIP, "sync failed");
    if (rc) { ohs_batch_destroy(b); return rc; }
    *out = b;
    return OHS_OK;
}

void ohs_batch_destroy(ohs_batch *b)
{
    if (!b) return;
    hipSetDevice(b->device);
    DeviceWideSection dws;
    hipDeviceSynchronize();
    conv_free(b->conv);
    if (b->eq.d_state) hipFree(b->eq.d_state);
    for (auto &sp : b->spans) { hipEventDestroy(sp.a); hipEventDestroy(sp.b); }
    for (hipEvent_t e : b->ev_pool) hipEventDestroy(e);
    for (hipEvent_t e : b->ev_inflight) hipEventDestroy(e);
    for (hipEvent_t e : b->chunk_done) hipEventDestroy(e);
    for (int k = 0; k < ohs_batch::kHostSlots; ++k) {
        if (b->d_slot[k]) hipFree(b->d_slot[k]);
        if (b->ev_h2d[k]) hipEventDestroy(b->ev_h2d[k]);
        if (b->ev_comp[k]) hipEventDestroy(b->ev_comp[k]);
        if (b->ev_d2h[k]) hipEventDestroy(b->ev_d2h[k]);
    }
    if (b->st_h2d) hipStreamDestroy(b->st_h2d);
    if (b->st_comp) hipStreamDestroy(b->st_comp);
    if (b->st_d2h) hipStreamDestroy(b->st_d2h);
    if (b->st2) hipStreamDestroy(b->st2);
    if (b->st) hipStreamDestroy(b->st);
    delete b;
}

int ohs_batch_set_ir(ohs_batch *b, int path, const float *ir, size_t len)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    HIP_TRY(hipSetDevice(b->device));
    DeviceWideSection dws;
    HIP_TRY(hipDeviceSynchronize());
    return conv_set_ir(b->conv, b->ctx, path, ir, len, b->st);
}

// node_batch.cpp: the same with the IR already on the batch's device (a broadcast buffer)
extern "C" int ohs_internal_batch_set_ir_device(ohs_batch *b, int path, const float *d_ir, size_t len)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    HIP_TRY(hipSetDevice(b->device));
    DeviceWideSection dws;
    HIP_TRY(hipDeviceSynchronize());
    return conv_set_ir(b->conv, b->ctx, path, d_ir, len, b->st, true);
}

int ohs_batch_set_eq_band_coeffs(ohs_batch *b, size_t band, const float coeffs[5], int enabled)
{
    if (!b || !coeffs) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    if (band >= b->eq.nb) return OHS_OK;
    std::memcpy(&b->eq.coeffs[5 * band], coeffs, 5 * sizeof(float));
    b->eq.en[band] = enabled != 0;
    return OHS_OK;
}

int ohs_batch_update_eq_band(ohs_batch *b, size_t band, float fs, int filter_type, float fc, float q,
                             float gain_db, int enabled)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    if (band >= b->eq.nb) return OHS_OK;
    float c[5];
    int rc = rbj(filter_type, fs, fc, q, gain_db, c);
    if (rc) return rc;
    std::memcpy(&b->eq.coeffs[5 * band], c, sizeof(c));
    b->eq.en[band] = enabled != 0;
    return OHS_OK;
}

int ohs_batch_set_eq_enabled(ohs_batch *b, int eq_enable)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    b->eq_enable = eq_enable != 0;
    return OHS_OK;
}

int ohs_batch_set_flush_denormals(ohs_batch *b, int mode)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    if (mode < 0 || mode > 2) return fail(OHS_ERR_INVALID_ARG, "mode must be 0 (IEEE), 1 (FTZ) or 2 (FTZ | DAZ)");
    b->conv.fp_mode = mode;
    b->eq.fp_mode = mode;
    return OHS_OK;
}

int ohs_batch_set_eq_exact_specials(ohs_batch *b, int enable)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    b->eq.exact_specials = enable != 0;
    return OHS_OK;
}

int ohs_batch_set_conv_plan(ohs_batch *b, int plan)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    if (plan < 0 || plan > 2) return fail(OHS_ERR_INVALID_ARG, "plan must be 0 (library's choice), 1 (block 512) or 2 (hop 1536)");
    b->conv.conv_plan = plan;
    return OHS_OK;
}

int ohs_batch_set_gain(ohs_batch *b, float gain)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    b->gain = gain;
    return OHS_OK;
}

int ohs_batch_reset(ohs_batch *b)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    HIP_TRY(hipSetDevice(b->device));
    DeviceWideSection dws;
    HIP_TRY(hipDeviceSynchronize());
    ConvState &c = b->conv;
    HIP_TRY(hipMemsetAsync(c.d_hist, 0, c.S * (size_t)c.cap * NF * sizeof(float2), b->st));
    HIP_TRY(hipMemsetAsync(c.d_tails, 0, c.S * 2 * 8 * 64 * sizeof(float2), b->st));
    c.tails_lazy = false;       // (the zeroed per-path overlaps are the state)
    c.cnt = 0;
    for (int p = 0; p < 4; ++p) c.since[p] = 0;
    int rc = eq_reset(b->eq, b->st);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->st));
    b->failed = false;          // (zeroed state is consistent state)
    b->fail_msg.clear();
    return OHS_OK;
}

static int batch_process_body(ohs_batch *b, const float *d_in, float *d_out, size_t n_blocks,
                              size_t stream_stride, size_t channel_stride, void *hip_stream, bool deferred);

static int batch_process_impl(ohs_batch *b, const float *d_in, float *d_out, size_t n_blocks,
                              size_t stream_stride, size_t channel_stride, void *hip_stream, bool deferred)
{
    if (!b || !d_in || !d_out) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    if (b->failed)
        return fail(OHS_ERR_HIP, "this batch failed in the middle of an earlier call (" + b->fail_msg +
                                     "): its per-stream state is half-advanced; ohs_batch_reset starts it afresh");
    const size_t spans_before = b->spans.size();
    const int rc = batch_process_body(b, d_in, d_out, n_blocks, stream_stride, channel_stride, hip_stream, deferred);
    if (rc == OHS_OK || rc == OHS_ERR_INVALID_ARG) return rc;      // (argument errors are found before anything is queued)
    // A HIP call failed with part of the work queued.  Keep the message, then leave nothing dangling:
    const std::string why = g_err;
    //  * the caller's stream must not run ahead of what this call put on the second stream
    hipStream_t st = (hipStream_t)hip_stream;
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
        if (hipEventRecord(ev, b->st2) != hipSuccess || hipStreamWaitEvent(st, ev, 0) != hipSuccess)
            hipStreamSynchronize(b->st2);
        b->ev_inflight.push_back(ev);
    } else {
        hipStreamSynchronize(b->st2);
    }
    b->join_pending = false;
    //  * timing spans opened by this call may hold events that were never recorded: drop them
    while (b->spans.size() > spans_before) {
        ohs_batch::Span sp = b->spans.back();
        b->spans.pop_back();
        if (sp.a) hipEventDestroy(sp.a);
        if (sp.b) hipEventDestroy(sp.b);
    }
    //  * some time chunks have advanced the per-stream state, others have not
    b->failed = true;
    b->fail_msg = why;
    return fail(rc, why);
}

static int batch_process_body(ohs_batch *b, const float *d_in, float *d_out, size_t n_blocks,
                              size_t stream_stride, size_t channel_stride, void *hip_stream, bool deferred)
{
    if (n_blocks == 0) return OHS_OK;
    if (n_blocks > (size_t)1 << 24) return fail(OHS_ERR_INVALID_ARG, "n_blocks too large");
    const size_t frames = n_blocks * BS;
    if (channel_stride < frames || (b->conv.S > 1 && stream_stride < 2 * frames && stream_stride < channel_stride + frames))
        return fail(OHS_ERR_INVALID_ARG, "strides smaller than the processed region");
    HIP_TRY(hipSetDevice(b->device));
    hipStream_t st = (hipStream_t)hip_stream;
    auto get_event = [&](hipEvent_t *e) -> int {
        if (!b->ev_pool.empty()) { *e = b->ev_pool.back(); b->ev_pool.pop_back(); return OHS_OK; }
        HIP_TRY(hipEventCreate(e));
        return OHS_OK;
    };
    // ordering events of earlier calls that have completed go back to the pool
    for (size_t i = 0; i < b->ev_inflight.size();) {
        if (hipEventQuery(b->ev_inflight[i]) == hipSuccess) {
            b->ev_pool.push_back(b->ev_inflight[i]);
            b->ev_inflight[i] = b->ev_inflight.back();
            b->ev_inflight.pop_back();
        } else ++i;
    }
    auto span_begin = [&](hipStream_t s_, int kind) -> int {
        if (!b->profiling) return OHS_OK;
        ohs_batch::Span sp; sp.kind = kind; sp.a = nullptr; sp.b = nullptr;
        int rc = get_event(&sp.a); if (rc) return rc;
        rc = get_event(&sp.b); if (rc) return rc;
        HIP_TRY(hipEventRecord(sp.a, s_));
        b->spans.push_back(sp);
        return OHS_OK;
    };
    auto span_end = [&](hipStream_t s_) -> int {
        if (!b->profiling) return OHS_OK;
        HIP_TRY(hipEventRecord(b->spans.back().b, s_));
        return OHS_OK;
    };
    if (b->profiling) b->prof_calls++;

    bool eq_active = false;
    if (b->eq_enable)
        for (size_t i = 0; i < b->eq.nb; ++i) eq_active = eq_active || b->eq.en[i];

    const long long ss = (long long)stream_stride, cs = (long long)channel_stride;
    int rc;
    // A deferred call may have left convolutions running on st2.  If this call repeats its geometry and
    // overlaps again, EQ chunk c only has to wait for THAT call's convolution of chunk c (same frames of
    // d_out); anything else joins completely first.
    const bool will_overlap = eq_active && b->overlap && n_blocks >= 64;
    bool chunk_waits = false;
    if (b->join_pending) {
        chunk_waits = will_overlap && b->pend_out == d_out && b->pend_blocks == n_blocks &&
                      b->pend_ss == stream_stride && b->pend_cs == channel_stride;
        if (!chunk_waits) {
            HIP_TRY(hipStreamWaitEvent(st, b->chunk_done[(size_t)b->chunk_done_n - 1], 0));
            b->join_pending = false;
        }
    }
    if (!eq_active) {     // lib.rs:1179 eq_enable false (or every band disabled: identity)
#ifdef OHS_EXPERIMENTS
        if (g_inject_batch_failure.load() > 0 && g_inject_batch_failure.fetch_sub(1) == 1)
            return fail(OHS_ERR_HIP, "injected failure (ohs_debug_inject_batch_failure)");
#endif
        rc = span_begin(st, 1); if (rc) return rc;
        rc = conv_launch(b->conv, b->ctx, d_in, ss, cs, d_out, ss, cs, (int)n_blocks, b->gain, st);
        if (rc) return rc;
        return span_end(st);
    }
    // The EQ is a serial recurrence (latency-bound, ~128 waves at 256 streams) and leaves most of the
    // chip idle, so the convolution of time chunk i runs on a second stream underneath the EQ of
    // chunk i+1.  Chunks touch disjoint frame ranges; state (EQ s1/s2, overlaps) chains per stream.
    // Uneven chunks: only the LAST chunk's convolution is not hidden under an EQ launch, so it is short
    // (2 % of the frames; six chunks measure 6.00-6.02 ms per headline step, the four of {0.34, 0.66, 0.92} 6.05,
    // seven 6.03-6.04: every extra EQ launch costs its ramp).
    const std::vector<double> &kCut = tuning().overlap_cuts;
    const int nch = (b->overlap && n_blocks >= 64) ? (int)kCut.size() - 1 : 1;
    // XCD partition of the overlapped launches: the EQ on XCDs [0, x), the convolution on [x, 8)
    struct XcdScope {
        ohs_batch *b;
        XcdScope(ohs_batch *b_, int x) : b(b_)
        {
            if (x > 0 && x < 8) { b->eq.xcd_lo = 0; b->eq.xcd_n = x; b->conv.xcd_lo = x; b->conv.xcd_n = 8 - x; }
        }
        ~XcdScope() { b->eq.xcd_lo = 0; b->eq.xcd_n = 8; b->conv.xcd_lo = 0; b->conv.xcd_n = 8; }
    } xcd_scope(b, nch > 1 ? b->xcd_split : 0);
    bool joined = false;        // the last convolution ran on the caller's stream: nothing left to join
    for (int i = 0; i < nch; ++i) {
        const size_t blk0 = nch == 1 ? 0 : (size_t)(kCut[i] * (double)n_blocks);
        const size_t blk1 = nch == 1 ? n_blocks : (i == nch - 1 ? n_blocks : (size_t)(kCut[i + 1] * (double)n_blocks));
        const size_t off = blk0 * BS;
        const int nb_i = (int)(blk1 - blk0);
        if (nb_i <= 0) {
            // two cut points truncated to the same block (OHS_OVERLAP_CUTS experiments): an empty chunk.
            // Its "done" event is recorded all the same so that a later deferred call's per-chunk wait
            // finds every event of this call on st2.
            if (nch > 1 && deferred) {
                while (b->chunk_done.size() <= (size_t)i) {
                    hipEvent_t ev;
                    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                    b->chunk_done.push_back(ev);
                }
                HIP_TRY(hipEventRecord(b->chunk_done[(size_t)i], b->st2));
            }
            continue;
        }
        if (chunk_waits) HIP_TRY(hipStreamWaitEvent(st, b->chunk_done[(size_t)i], 0));
#ifdef OHS_EXPERIMENTS
        if (g_inject_batch_failure.load() > 0 && g_inject_batch_failure.fetch_sub(1) == 1)
            return fail(OHS_ERR_HIP, "injected failure (ohs_debug_inject_batch_failure)");
#endif
        // The EQ launch carries its own events (start / completion of the dispatch: no marker packets between the
        // back-to-back EQ launches of a step); the completion event is what the convolution's stream waits for.
        hipEvent_t ev_a = nullptr, ev_b = nullptr;
        if (b->profiling) {
            ohs_batch::Span sp; sp.kind = 0; sp.a = nullptr; sp.b = nullptr;
            rc = get_event(&sp.a); if (rc) return rc;
            rc = get_event(&sp.b); if (rc) return rc;
            b->spans.push_back(sp);
            ev_a = sp.a; ev_b = sp.b;
        } else if (nch > 1) {
            rc = get_event(&ev_b); if (rc) return rc;
            b->ev_inflight.push_back(ev_b);
        }
        rc = eq_launch(b->eq, d_in + off, d_out + off, ss, cs, (long long)nb_i * BS, st, nullptr, ev_a, ev_b);
        if (rc) return rc;
        hipStream_t cst = st;
        // The LAST chunk's convolution has nothing to hide under: it runs on the caller's stream right behind its EQ
        // launch (after the earlier convolutions on st2, whose state it continues), which spares the step the hop to
        // st2 and the join back -- two cross-stream waits of 10-25 us each in front of the caller's next launch.
        const bool tail_on_caller = nch > 1 && !deferred && i == nch - 1;
        if (tail_on_caller) {
            hipEvent_t ev;
            rc = get_event(&ev); if (rc) return rc;
            HIP_TRY(hipEventRecord(ev, b->st2));
            HIP_TRY(hipStreamWaitEvent(st, ev, 0));
            b->ev_inflight.push_back(ev);
            joined = true;
        } else if (nch > 1) {
            HIP_TRY(hipStreamWaitEvent(b->st2, ev_b, 0));
            cst = b->st2;
        }
        {
            hipEvent_t cv_a = nullptr, cv_b = nullptr;
            if (b->profiling) {
                ohs_batch::Span sp; sp.kind = 1; sp.a = nullptr; sp.b = nullptr;
                rc = get_event(&sp.a); if (rc) return rc;
                rc = get_event(&sp.b); if (rc) return rc;
                b->spans.push_back(sp);
                cv_a = sp.a; cv_b = sp.b;
            }
            rc = conv_launch(b->conv, b->ctx, d_out + off, ss, cs, d_out + off, ss, cs, nb_i, b->gain, cst, true, cv_a, cv_b);
            if (rc) return rc;
        }
        if (nch > 1 && deferred) {
            while (b->chunk_done.size() <= (size_t)i) {
                hipEvent_t ev;
                HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                b->chunk_done.push_back(ev);
            }
            HIP_TRY(hipEventRecord(b->chunk_done[(size_t)i], b->st2));
        }
    }
    b->join_pending = false;
    if (nch > 1 && deferred) {      // the caller joins later (ohs_batch_join / ohs_batch_sync / the next call)
        b->join_pending = true;
        b->chunk_done_n = nch;
        b->pend_out = d_out; b->pend_blocks = n_blocks; b->pend_ss = stream_stride; b->pend_cs = channel_stride;
    } else if (nch > 1 && !joined) {    // join: the caller's stream continues only after the last convolution
        hipEvent_t ev;
        rc = get_event(&ev); if (rc) return rc;
        HIP_TRY(hipEventRecord(ev, b->st2));
        HIP_TRY(hipStreamWaitEvent(st, ev, 0));
        b->ev_inflight.push_back(ev);
    }
    return OHS_OK;
}

int ohs_batch_process(ohs_batch *b, const float *d_in, float *d_out, size_t n_blocks,
                      size_t stream_stride, size_t channel_stride, void *hip_stream)
{
    return batch_process_impl(b, d_in, d_out, n_blocks, stream_stride, channel_stride, hip_stream, false);
}

int ohs_batch_process_deferred(ohs_batch *b, const float *d_in, float *d_out, size_t n_blocks,
                               size_t stream_stride, size_t channel_stride, void *hip_stream)
{
    return batch_process_impl(b, d_in, d_out, n_blocks, stream_stride, channel_stride, hip_stream, true);
}

// Host-buffer batch call (north_star's offline mode fed from host memory).  The frames are cut into time
// chunks; chunk i + 1 is copied in, chunk i processed and chunk i - 1 copied out at the same time, on three
// streams over three device staging slots, so that both directions of the host link and the kernels
// overlap (the 8 B/frame in + 8 B/frame out over PCIe, not the GPU work, is the bound: DESIGN.md section 5).
// Each chunk is one ohs_batch_process call on its slot, in stream order, so per-stream state chains exactly
// as in a sequence of device calls with the same chunk sizes.
int ohs_batch_process_host(ohs_batch *b, const float *h_in, float *h_out, size_t n_blocks,
                           size_t stream_stride, size_t channel_stride, size_t chunk_blocks)
{
    if (!b || !h_in || !h_out) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    if (n_blocks == 0) return OHS_OK;
    if (n_blocks > (size_t)1 << 24) return fail(OHS_ERR_INVALID_ARG, "n_blocks too large");
    const size_t frames = n_blocks * BS, S = b->conv.S;
    if (channel_stride < frames || (S > 1 && stream_stride < 2 * frames && stream_stride < channel_stride + frames))
        return fail(OHS_ERR_INVALID_ARG, "strides smaller than the processed region");
    HIP_TRY(hipSetDevice(b->device));
    if (chunk_blocks == 0) {
        // default: ~32 MiB per chunk and direction (0.7 ms of link time; 16 ... 64 MiB measure alike, smaller chunks pay
        // per-copy overhead, larger ones a longer fill and drain), at most an eighth of the call so that small batches
        // pipeline too, at least 16 blocks
        const size_t per_block = S * 2 * BS * sizeof(float);
        chunk_blocks = ((size_t)32 << 20) / per_block;
        chunk_blocks = std::min(chunk_blocks, std::max<size_t>(n_blocks / 8, 1));
        chunk_blocks = std::max<size_t>(chunk_blocks, 16);
    }
    chunk_blocks = std::min(chunk_blocks, n_blocks);
    const size_t cf = chunk_blocks * BS;
    if (!b->st_h2d) {
        // The runtime multiplexes streams onto a few hardware queues per PRIORITY level (4 by default,
        // GPU_MAX_HW_QUEUES), and a stream that shares its queue with a copy stream waits behind every chunk
        // copy: with plain streams the convolution launches of chunk i + 1 sat behind the copy-out of chunk i
        // (rocprofv3 --memory-copy-trace: 2.6 ms per chunk instead of 2.1).  The two copy streams therefore get
        // priority levels of their own, where nothing else of this process lives.
        int pr_least = 0, pr_greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
        if (tuning().host_pipe_flat_priorities) pr_least = pr_greatest = 0;     // (experiments: the old behaviour)
        // (the pipeline's compute stream too: which normal-priority stream it would share a queue with depends on how
        // many streams the process has created before -- bench.py measured 21.7 or 25.6 ms depending on its step count)
        HIP_TRY(hipStreamCreateWithPriority(&b->st_h2d, hipStreamNonBlocking, pr_greatest));
        HIP_TRY(hipStreamCreateWithPriority(&b->st_comp, hipStreamNonBlocking, pr_greatest));
        HIP_TRY(hipStreamCreateWithPriority(&b->st_d2h, hipStreamNonBlocking, pr_least != 0 ? pr_least : pr_greatest));
        for (int k = 0; k < ohs_batch::kHostSlots; ++k) {
            HIP_TRY(hipEventCreateWithFlags(&b->ev_h2d[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&b->ev_comp[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&b->ev_d2h[k], hipEventDisableTiming));
        }
    }
    if (cf > b->slot_frames) {
        DeviceWideSection dws;
        HIP_TRY(hipDeviceSynchronize());
        for (int k = 0; k < ohs_batch::kHostSlots; ++k) {
            if (b->d_slot[k]) hipFree(b->d_slot[k]);
            b->d_slot[k] = nullptr;
        }
        b->slot_frames = 0;
        for (int k = 0; k < ohs_batch::kHostSlots; ++k) HIP_TRY(hipMalloc(&b->d_slot[k], S * 2 * cf * sizeof(float)));
        b->slot_frames = cf;
    }
    // one 2-D copy per chunk when the host rows are equally spaced ([stream][channel] with stream_stride ==
    // 2 * channel_stride, or a single stream); else one 2-D copy per stream
    const bool regular = S == 1 || stream_stride == 2 * channel_stride;
    auto copy = [&](bool to_device, float *dev, size_t off, size_t nf, hipStream_t st) -> int {
        const size_t dpitch = b->slot_frames * sizeof(float), w = nf * sizeof(float);
        if (regular) {
            const size_t hpitch = channel_stride * sizeof(float);
            if (to_device)
                HIP_TRY(hipMemcpy2DAsync(dev, dpitch, h_in + off, hpitch, w, S * 2, hipMemcpyHostToDevice, st));
            else
                HIP_TRY(hipMemcpy2DAsync(h_out + off, hpitch, dev, dpitch, w, S * 2, hipMemcpyDeviceToHost, st));
            return OHS_OK;
        }
        for (size_t s_ = 0; s_ < S; ++s_) {
            const size_t hpitch = channel_stride * sizeof(float);
            float *d = dev + s_ * 2 * b->slot_frames;
            if (to_device)
                HIP_TRY(hipMemcpy2DAsync(d, dpitch, h_in + s_ * stream_stride + off, hpitch, w, 2, hipMemcpyHostToDevice, st));
            else
                HIP_TRY(hipMemcpy2DAsync(h_out + s_ * stream_stride + off, hpitch, d, dpitch, w, 2, hipMemcpyDeviceToHost, st));
        }
        return OHS_OK;
    };
    const size_t n_chunks = (n_blocks + chunk_blocks - 1) / chunk_blocks;
    int rc = OHS_OK;
    // Work the caller queued earlier through ohs_batch_process* touches the same per-stream state.  A pending
    // deferred call is joined here (its convolutions run on the handle's own second stream); work on the CALLER's
    // streams cannot be seen from here: the header asks for ohs_batch_sync first.
    if (b->join_pending) {
        HIP_TRY(hipStreamWaitEvent(b->st_comp, b->chunk_done[(size_t)b->chunk_done_n - 1], 0));
        b->join_pending = false;
    }
    // (experiments build, host_pipe_trace: device timestamps around every stage of every chunk, printed after the call;
    // it adds six event records per chunk)
    const bool trace = tuning().host_pipe_trace != 0;
    std::vector<hipEvent_t> tev;
    auto mark = [&](hipStream_t st) {
        if (!trace) return;
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) == hipSuccess) { hipEventRecord(e, st); tev.push_back(e); }
    };
    const auto host_t0 = std::chrono::steady_clock::now();
    std::vector<double> host_ms;
    // inside the loop a HIP failure must not return: the drain below has to run (the copies already queued write
    // into the caller's buffers)
#define PIPE_TRY(x)                                                                                        \
    {                                                                                                      \
        const hipError_t pe_ = (x);                                                                        \
        if (pe_ != hipSuccess) { rc = fail(OHS_ERR_HIP, std::string(#x ": ") + hipGetErrorString(pe_)); break; } \
    }
    for (size_t i = 0; i < n_chunks && rc == OHS_OK; ++i) {
        const int k = (int)(i % ohs_batch::kHostSlots);
        const size_t blk0 = i * chunk_blocks, nb = std::min(chunk_blocks, n_blocks - blk0);
        const size_t off = blk0 * BS, nf = nb * BS;
        if (i >= (size_t)ohs_batch::kHostSlots) PIPE_TRY(hipStreamWaitEvent(b->st_h2d, b->ev_d2h[k], 0))    // slot free again
        mark(b->st_h2d);
        rc = copy(true, b->d_slot[k], off, nf, b->st_h2d);
        if (rc) break;
        mark(b->st_h2d);
        PIPE_TRY(hipEventRecord(b->ev_h2d[k], b->st_h2d))
        PIPE_TRY(hipStreamWaitEvent(b->st_comp, b->ev_h2d[k], 0))
        mark(b->st_comp);
        rc = batch_process_impl(b, b->d_slot[k], b->d_slot[k], nb, 2 * b->slot_frames, b->slot_frames, b->st_comp, false);
        if (rc) break;
        mark(b->st_comp);
        PIPE_TRY(hipEventRecord(b->ev_comp[k], b->st_comp))
        PIPE_TRY(hipStreamWaitEvent(b->st_d2h, b->ev_comp[k], 0))
        mark(b->st_d2h);
        rc = copy(false, b->d_slot[k], off, nf, b->st_d2h);
        if (rc) break;
        mark(b->st_d2h);
        PIPE_TRY(hipEventRecord(b->ev_d2h[k], b->st_d2h))
        if (trace) host_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count());
    }
#undef PIPE_TRY
    // blocking call: the outputs are complete on return (also on a failure half-way: nothing may still be
    // writing into the caller's buffers)
    const hipError_t e1 = hipStreamSynchronize(b->st_h2d), e2 = hipStreamSynchronize(b->st_comp),
                     e3 = hipStreamSynchronize(b->st_d2h);
    if (trace && !tev.empty()) {
        fprintf(stderr, "[ohs host pipeline] %zu chunks of %zu blocks; per chunk [ms since the first copy began]: "
                        "h2d begin-end | kernels begin-end | d2h begin-end | host enqueued at\n", n_chunks, chunk_blocks);
        for (size_t i = 0; i + 5 < tev.size(); i += 6) {
            float t[6];
            for (int j = 0; j < 6; ++j) if (hipEventElapsedTime(&t[j], tev[0], tev[i + j]) != hipSuccess) t[j] = -1.f;
            fprintf(stderr, "  %2zu: %6.2f-%6.2f | %6.2f-%6.2f | %6.2f-%6.2f | %6.2f\n", i / 6, t[0], t[1], t[2], t[3], t[4], t[5],
                    i / 6 < host_ms.size() ? host_ms[i / 6] : -1.0);
        }
        for (hipEvent_t e : tev) hipEventDestroy(e);
    }
    if (rc) return rc;
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(OHS_ERR_HIP, "host pipeline: stream sync failed");
    return OHS_OK;
}

int ohs_batch_join(ohs_batch *b, void *hip_stream)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    if (!b->join_pending) return OHS_OK;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)hip_stream, b->chunk_done[(size_t)b->chunk_done_n - 1], 0));
    b->join_pending = false;
    return OHS_OK;
}

int ohs_batch_set_profiling(ohs_batch *b, int enable)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    b->profiling = enable != 0;
    return OHS_OK;
}

int ohs_batch_profile_read(ohs_batch *b, double *eq_ms, double *conv_ms, uint64_t *n_calls,
                           uint64_t *eq_launches, uint64_t *conv_launches)
{
    if (!b || !eq_ms || !conv_ms || !n_calls || !eq_launches || !conv_launches)
        return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(b->device));
    double t[2] = {0.0, 0.0};
    uint64_t cntk[2] = {0, 0};
    for (auto &sp : b->spans) {
        cntk[sp.kind]++;
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(sp.b));
        HIP_TRY(hipEventElapsedTime(&ms, sp.a, sp.b));
        t[sp.kind] += ms;
        b->ev_pool.push_back(sp.a);
        b->ev_pool.push_back(sp.b);
    }
    b->spans.clear();
    *eq_ms = t[0]; *conv_ms = t[1]; *n_calls = b->prof_calls;
    *eq_launches = cntk[0]; *conv_launches = cntk[1];
    b->prof_calls = 0;
    return OHS_OK;
}

int ohs_batch_sync(ohs_batch *b, void *hip_stream)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    if (b->join_pending) {
        HIP_TRY(hipStreamSynchronize(b->st2));
        b->join_pending = false;
    }
    return OHS_OK;
}

int ohs_batch_algorithmic_bytes(const ohs_batch *b, size_t n_blocks, uint64_t *bytes)
{
    if (!b || !bytes) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    // SURVEY.md section 8d / DESIGN.md byte model, per 512-frame block per stream:
    //   audio in 4096 + out 4096, history write 2*4104, history read 2*(P-1)*4104,
    //   overlap read+write 2*(2048+2048), EQ state + coefficients ~520 (if EQ on),
    //   HRIR spectra 4*P*4104 once per block-time, shared by all streams.
    int P = 1;
    for (int p = 0; p < 4; ++p) P = std::max(P, b->conv.P[p]);
    const uint64_t per_stream_block = 4096u + 4096u + 2u * 4104u + 2u * (uint64_t)(P - 1) * 4104u +
                                      2u * (2048u + 2048u) + (b->eq_enable ? 520u : 0u);
    const uint64_t shared_per_block = 4u * (uint64_t)P * 4104u;
    *bytes = (uint64_t)n_blocks * ((uint64_t)b->conv.S * per_stream_block + shared_per_block);
    return OHS_OK;
}

int ohs_batch_kernel_bytes(const ohs_batch *b, size_t n_blocks, uint64_t *eq_bytes, uint64_t *conv_bytes)
{
    if (!b || !eq_bytes || !conv_bytes) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    int P = 1;
    for (int p = 0; p < 4; ++p) P = std::max(P, b->conv.P[p]);
    const uint64_t S = b->conv.S, nbk = n_blocks;
    // DESIGN bytes: what the kernels are built to move through HBM for one call of n_blocks (tables and ring
    // re-reads that are served by L2 are not counted).  This is NOT the SURVEY 8d per-block model
    // (ohs_batch_algorithmic_bytes): the P = 1 kernel keeps overlaps in registers and writes no history.
    //   EQ:             audio in + out, state + coefficients
    //   P = 1:          audio in + out; per stream and chunk boundary the input block in front of it once more (4 KiB;
    //                   where the pre-pass computes the boundary tails, K not in {2, 4, 8, 16}: the tail written and read
    //                   as well) and the four-overlap state (8 KiB in + 8 KiB out)
    //   P > 1 (time-parallel): audio in, ring write, ring read once, W write, W read, audio out
    const uint64_t eq_sb = 4096u + 4096u + 520u;
    *eq_bytes = b->eq_enable ? nbk * S * eq_sb : 0;
    if (P == 1) {
        const uint64_t K = (uint64_t)conv_p1_chunks(b->ctx, (size_t)S, (long long)nbk, 0);
        const bool own_tails = conv_p1_waves_per_cu() == 16 && (K == 2 || K == 4 || K == 8 || K == 16);
        // state: per-path overlaps 8 KiB in + 8 KiB out, or (lazy, kernels.h) merged overlap 4 KiB in + 4 KiB out + the
        // last block's input copy 4 KiB
        const uint64_t state = b->conv.lazy_ok ? 3u * 4096u : 2u * 8192u;
        *conv_bytes = nbk * S * 8192u + S * ((K - 1) * (own_tails ? 1u : 3u) * 4096u + state);
    } else {
        *conv_bytes = nbk * S * (4096u + 8192u + 8192u + 8192u + 8192u + 4096u);
    }
    return OHS_OK;
}

}  // extern "C"
