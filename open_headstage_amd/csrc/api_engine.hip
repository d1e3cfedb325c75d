// api_engine.hip -- ohs_engine_* (ConvolutionEngine, convolution.rs:68-295): the host-block FIFO adapter on the
// device, the launch-per-call path, the resident real-time kernel with its pinned mailbox and the per-device registry
// that keeps device-wide waits from stalling on it; ohs_chain_process (lib.rs:1169-1207).
#include "api_internal.h"

using namespace ohs;
using namespace ohs_api;

namespace {


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

// The buffer of the sums computed ahead (kernels.h: ConvGeneralArgs::pre) exists from the first response of more than one
// partition on -- allocated where set_ir / clone allocate, never on the audio thread.
int engine_pre_reserve(ConvState &c)
{
    if (c.d_pre || conv_max_p(c) <= 1) return OHS_OK;
    HIP_TRY(hipMalloc(&c.d_pre, (size_t)8 * 2 * NF * sizeof(float2)));
    c.pre_cnt = ~0ull; c.pre_n = 0;
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
    conv_pre_args(c, a, (int)k);
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
    conv_pre_served(c, (int)k);
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
    const int conf = rt_max_resident_load();
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
}  // namespace
namespace ohs_api {
void rt_request_stop_locked(ohs_engine *e)       // registry mutex held: e->mb is alive (declared in api_internal.h: DeviceWideSection)
{
    if (e->mb) __atomic_store_n(&e->mb->call_word[3], 1u, __ATOMIC_RELEASE);
}
}  // namespace ohs_api
namespace {

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
            ohsint_set_error("resident kernel did not leave within 5 s of its stop request; the engine is unusable until it has");
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
    conv_pre_args(c, a, 0);     // (the kernel sets pre_n itself after every call that completes a block; the host mirrors it)
    EngineCallArgs &g = mb->call;
    g.in_fifo = e->in.d; g.in_cap = (int)e->in.cap; g.in_len = (int)e->in.len;
    g.out_fifo = e->out.d; g.out_cap = (int)e->out.cap; g.out_len = (int)e->out.len;
    g.src_l = e->dm_in; g.src_r = e->dm_in + e->hcap;
    g.host_l = e->dm_out; g.host_r = e->dm_out + e->hcap;
    g.n = 0; g.emit = 0; g.gain_pop = 1.0f;
    mb->call_word[0] = e->rt_seq; mb->call_word[1] = 0; mb->call_word[2] = 0; mb->call_word[3] = 0;
    mb->done = e->rt_seq; mb->alive = 1;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    const unsigned long long idle_limit = rt_idle_polls_load();   // two reads in flight: one poll per ~0.6 us
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
    conv_pre_served(c, (int)k);
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
    // Host blocks of 4 097 .. 8 192 frames (a DAW's largest buffers): the path below stages through four stream copies
    // from and to the caller's pageable buffers and costs ~100 us before its first frame (an 8 192-frame call took 3.5 x a
    // 4 096-frame one: profiles/r05_engine_call_sizes_before.jsonl); two calls of the one-launch kernel serve such a block
    // in about half of that while a block needs few partition products.  Same result as ONE call (convolution.rs:141-182):
    // the first part completes exactly 8 blocks and empties the input FIFO; if the whole call pops its n frames
    // (out_len + 512 k >= n, :163), so does each part -- out_len + 4096 >= n1 always, and the second part's condition
    // is the whole call's with n1 subtracted on both sides.  A call that would answer with silence (:176-181) keeps the
    // path below.  The chain's EQ is sample-serial: splitting its block changes nothing.
    if (n <= 8192 && e->in.len < (size_t)BS && conv_max_p(e->conv) <= 8) {
        const size_t k = (e->in.len + n) / BS;
        if (e->out.len + k * BS >= n) {
            const size_t n1 = (size_t)8 * BS - e->in.len;
            const int rc1 = engine_process_single_launch(e, in_l, in_r, out_l, out_r, n1, eq, gain);
            if (rc1) return rc1;
            return engine_process_single_launch(e, in_l + n1, in_r + n1, out_l + n1, out_r + n1, n - n1, eq, gain);
        }
    }
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

extern "C" {

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
        ohsint_set_error("ohs_engine_destroy: a resident kernel never left; its mailbox, stream and device state were leaked");
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
    if (engine_pre_reserve(d) != OHS_OK) return bail(OHS_ERR_HIP);      // (its first call computes every partition itself)
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
    rc = conv_grow_ring(e->conv, maxP + 7, e->st);
    if (rc) return rc;
    return engine_pre_reserve(e->conv);
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
        e->conv.pre_cnt = ~0ull;    // (sums computed ahead were computed in the old mode)
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

int ohs_engine_set_speakers(ohs_engine *e, const ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r,
                            float radius_m, float fs, unsigned *changed_mask)
{
    if (!e) return fail(OHS_ERR_INVALID_ARG, "engine is NULL");
    return set_speakers_impl(e->conv, sofa, az_l, el_l, az_r, el_r, radius_m, fs, changed_mask,
                             [&](int p, const float *ir, size_t n) { return ohs_engine_set_ir(e, p, ir, n); });
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

}  // extern "C"
