// api_internal.h -- what the translation units of the C-ABI implementation share: error reporting, the per-device
// context, the convolution / EQ state of a handle with the functions that drive it, the handle structs themselves.
// Internal to open_headstage_amd/csrc; nothing here crosses the C ABI (include/ohs_hip.h).
//   api_core.hip    error state, device context, library-wide entry points (version, device count, ohs_rt_configure, debug)
//   api_conv.hip    ConvState: set_ir, the launch plans of the convolution (conv_launch)
//   api_eq.hip      EqState + ohs_eq_* + ohs_biquad_*
//   api_engine.hip  ohs_engine_* (FIFO adapter, launch per call, resident real-time kernel and its mailbox), ohs_chain_process
//   api_batch.hip   ohs_batch_* (EQ || convolution over time chunks, host pipeline, profiling)
#pragma once
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

namespace ohs_api {
using namespace ohs;

constexpr int BS = OHS_BLOCK_SIZE;
constexpr int NF = OHS_FFT_SIZE;

extern thread_local std::string g_err;
#ifdef OHS_EXPERIMENTS
// experiments build: the k-th chunk launch sequence of ohs_batch_process from now fails as if HIP had refused it
// (ohs_debug_inject_batch_failure); the product build has no such path
extern std::atomic<int> g_inject_batch_failure;
#endif

int fail(int status, const std::string &msg);

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
    float2 *d_tw4096 = nullptr; // k_conv_lb_*: w4096^(n1 c), [3][1024]
    float2 *d_tw16384 = nullptr;// k_conv_xb: w16384^(n1 c), [15][1024]
    int *d_pairs = nullptr;     // k_conv_tp_mac: leaders of the spectrum position pairs {b, mirror(b)}
    int n_pairs = 0;
    int num_cus = 256;
};
int get_ctx(int device, DeviceCtx **out);

// the MAC reads up to Ppad - 1 blocks back (zero C/D beyond the real partition count)
inline int Ppad_for_ring(int Pmax) { return (Pmax + 15) / 16 * 16; }

inline int next_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

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
extern RtRegistry g_rt_reg[kMaxDevices];
void rt_request_stop_locked(ohs_engine *e);      // (api_engine.hip, with the mailbox protocol)
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

// ---- convolution state shared by the engine (1 stream) and batch (S streams); api_conv.hip ----
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
    int conv_plan = 0;              // ohs_batch_set_conv_plan: 0 = the library's choice, 1 = block 512 / FFT 1024, 2 = the large-transform
                                    // plans (one partition: hop 1536 / FFT 2048; longer responses: block 2048 / FFT 4096)
    // Block-2048 plan for long impulse responses (conv_lb_kernels.hip; batch objects with a response of lb_min_p() partitions
    // or more).  It is overlap-save on the INPUT: the state it needs and leaves is the last xh_len input frames of every
    // stream (d_xhist; every P > 1 launch of such an object keeps it current), not spectra and overlaps.  While lb_lazy,
    // d_hist / d_tails are stale; conv_materialise_state rebuilds them from d_xhist when another plan needs them.
    float *d_irl = nullptr;         // [4][irl_len] the impulse responses in the time domain, zero-padded
    long long irl_len = 0;
    float *d_xhist = nullptr;       // [S][2][2 xh_len], a ring of twice its reach: the frame k <= xh_len frames in front of the next
                                    // one at (xh_head - k) mod 2 xh_len (the forward kernels append into the half nobody reads)
    float *d_xhist_alt = nullptr;   // [S][2][xh_len] scratch: the history in time order, when the block-512 state is rebuilt from it
    long long xh_len = 0;           // the reach = P2pad * 2048
    long long xh_head = 0;          // where the next frame goes (a multiple of 512, < 2 xh_len)
    long long xh_valid = 0;         // trailing frames of d_xhist that are what every path may see of the past (zeros after a reset of all paths count)
    bool lb_lazy = false;
    float4 *d_lb_ring = nullptr;    // [S][lb_ring_cap][2048] window spectra, pair layout
    int lb_ring_cap = 0;
    // The windows sit on the stream's ABSOLUTE grid (block B = frames [2048 B, 2048 B + 2048) counted from the last reset):
    // window B lives in ring slot (B + 4 lb_ring_cap) % lb_ring_cap.  A call may start and end anywhere on the 512 grid: the
    // block it starts in is recomputed (its earlier frames come from the input history), the block it ends in is computed
    // with zeros for the frames to come and recomputed by the next call -- only windows of COMPLETE blocks stay valid.
    long long lb_end = 0;           // windows lb_end - lb_valid .. lb_end - 1 (complete blocks) are in the ring
    long long lb_valid = 0;
    float4 *d_lb_cd = nullptr, *d_lb_ab = nullptr;          // [P2pad][2][2048] tables; [2][P2pad][2048] scratch for building them
    int lb_p2pad = 0;               // what d_lb_cd / d_lb_ab are allocated for
    bool lb_cd_valid = false;
    // Pending tails (conv_lb_tail_route): a per-path set_ir on an object that keeps the input history is carried out as "every
    // path forgets its past" + what the frames in front of the reset still owe the output through the SURVIVING paths,
    // computed once ([S][4 paths][pt_len], a row per path so that a later reset of that path can drop it) and added to the
    // output of the calls that follow, whatever plan serves them.  No kernel of such an object ever needs a per-path gate.
    float *d_ptail = nullptr, *d_ptail_alt = nullptr;
    long long pt_len = 0, pt_pos = 0;
    bool pt_active = false;
    float4 *d_lb_cd_alt = nullptr;  // tables of one speaker pair alone (the tails' two runs)
    int lb_cd_alt_p2pad = 0;
    // block 8192 / FFT 16384 (conv_xb_kernels.hip): tables of the response's one or two 8192-tap partitions; valid with lb_cd_valid's
    // generation (every place that invalidates the block-2048 tables bumps lb_tables_gen)
    float4 *d_xb_cd = nullptr, *d_xb_ab = nullptr;
    int xb_p2 = 0, xb_form = 0;
    unsigned long long lb_tables_gen = 1, xb_cd_gen = 0;
    // Engines (one stream) with a response of more than one partition: the next call's sums over the partitions behind it
    // (kernels.h: ConvGeneralArgs::pre), computed by the kernel that served the previous call.  They belong to the state
    // with block counter pre_cnt; anything else that touches the state (set_ir, reset, a ring that moved, another kernel
    // that processes blocks) leaves pre_cnt != cnt or sets it to ~0 here.
    float2 *d_pre = nullptr;        // [8][2][1024]
    unsigned long long pre_cnt = ~0ull;
    int pre_n = 0;
    // what served the last convolution launch (ohs_batch_last_conv_plan): OHS_CONV_KERNEL_*, and its ranges per stream
    int last_kernel = 0, last_ranges = 0;
    unsigned long long kernel_calls[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // conv_launch calls served per family (ohs_batch_conv_plan_counts)
    // ohs_*_set_speakers: what it last loaded into each path (a set_ir from anywhere else forgets it), so that a
    // change of the speaker angles re-loads only the paths whose impulse response really changed
    std::vector<float> spk_ir[4];
    bool spk_set[4] = {false, false, false, false};
};

void conv_free(ConvState &c);
inline int conv_max_p(const ConvState &c) { return std::max(std::max(c.P[0], c.P[1]), std::max(c.P[2], c.P[3])); }
// the `pre` fields of an engine kernel's arguments for a call that completes k blocks (<= 8)
inline void conv_pre_args(const ConvState &c, ConvGeneralArgs &a, int k)
{
    const bool on = c.d_pre != nullptr && conv_max_p(c) > 1;
    a.pre = on ? c.d_pre : nullptr;
    a.pre_n = (on && c.pre_cnt == c.cnt) ? c.pre_n : 0;
    a.pre_next = on ? k : 0;
}
// ... and the host's mirror of what the kernel left behind (after c.cnt has been advanced by k)
inline void conv_pre_served(ConvState &c, int k)
{
    if (k <= 0) return;             // no block completed: the sums found stay what they are
    const bool on = c.d_pre != nullptr && conv_max_p(c) > 1;
    c.pre_cnt = on ? c.cnt : ~0ull;
    c.pre_n = on ? k : 0;
}
int conv_init(ConvState &c, size_t S, hipStream_t st);
int conv_enable_lazy_state(ConvState &c);
int conv_materialise_state(ConvState &c, DeviceCtx *ctx, hipStream_t st);
int conv_grow_ring(ConvState &c, int need, hipStream_t st);
int conv_set_ir(ConvState &c, DeviceCtx *ctx, int path, const float *ir, size_t len, hipStream_t st,
                bool ir_on_device = false);
bool conv_plan_auto_is_os(size_t S, long long n_blocks, bool in_place);
inline int lb_min_p() { return tuning().lb_min_p; }     // 512-tap partitions from which the block-2048 plan is considered (2: taps > 512)
bool conv_plan_auto_is_lb(size_t S, long long n_blocks, int Pmax);
bool conv_plan_auto_is_xb(size_t S, long long n_blocks, int Pmax);      // ... and among those, the block-8192 kernel (out-of-place calls)
int conv_os_chunks(const DeviceCtx *ctx, size_t S, long long n_blocks, bool in_place);
// time chunks per stream of the one-partition kernel's launch plan (also what ohs_batch_kernel_bytes prices)
long long conv_p1_chunks(const DeviceCtx *ctx, size_t S, long long n_blocks, long long target_override, int xcd_n = 8);
int conv_launch(ConvState &c, DeviceCtx *ctx, const float *in, long long in_ss, long long in_cs,
                float *out, long long out_ss, long long out_cs, int n_blocks, float gain, hipStream_t st,
                bool allow_fast = true, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// ---- speaker angles -> four set_ir (the wiring the reference leaves open; speakers.cpp) ---------
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
    // Per-stream tables (ohs_batch_set_stream_eq_band_coeffs): every stream owns its bands, as every plugin instance of the
    // reference does (parametric_eq.rs:125-129).  Host copy [streams][nb] of coefficients and enable flags, created from the
    // shared table by the first per-stream edit; the compacted device tables are uploaded by the next launch.
    bool per_stream = false;
    std::vector<float> s_coeffs;         // streams * nb * 5
    std::vector<unsigned char> s_en;     // streams * nb
    EqStreamTable *d_stabs = nullptr;    // [passes][streams]: pass k = every stream's enabled bands 12 k .. 12 k + 11
    size_t stabs_passes = 0;
    bool stabs_dirty = true;
    size_t max_enabled = 0;              // largest number of enabled bands of any stream (as of the last upload scan)
    unsigned long long *d_stamps = nullptr;     // [4] clock stamps of the last ring launch's wave 0 (ohs_batch_set_profiling)
};


int eq_init(EqState &e, size_t nb, size_t chains, float fs, hipStream_t st);
int eq_reset(EqState &e, hipStream_t st);
int eq_single_pass_table(const EqState &e, EqPassTable &t);
bool eq_any_enabled(const EqState &e);      // some band of the shared table, or of any stream's own table, is enabled
int eq_set_stream_band(EqState &e, size_t stream, size_t band, const float coeffs[5], int enabled);
void eq_set_shared_band(EqState &e, size_t band, const float coeffs[5], int enabled);   // ... and of every stream's own table
void eq_share_table(EqState &e);            // back to the one shared table
void eq_free(EqState &e);
int eq_launch(EqState &e, const float *in, float *out, long long ss, long long cs, long long n,
              hipStream_t st, bool *did_anything = nullptr, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// ---- growable planar device FIFO [2][cap] (api_engine.hip) -----------------------------
struct DevFifo {
    float *d = nullptr;
    size_t cap = 0, len = 0;
};

}  // namespace ohs_api

using ohs_api::ConvState;
using ohs_api::DevFifo;
using ohs_api::DeviceCtx;
using ohs_api::EqState;
using ohs::RtMailbox;

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
