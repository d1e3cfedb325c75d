// api_batch.hip -- ohs_batch_*: S independent streams sharing the four impulse responses and the EQ table (the offline
// many-stream job of BASELINE.json's north_star): EQ || convolution over time chunks on two streams, the deferred form,
// the PCIe-fed host pipeline, per-kernel profiling, fail-closed error handling.
#include "api_internal.h"

using namespace ohs;
using namespace ohs_api;
using ohs_host::rbj;

extern "C" {

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
    eq_free(b->eq);
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
extern "C" int ohsint_batch_set_ir_device(ohs_batch *b, int path, const float *d_ir, size_t len)
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
    eq_set_shared_band(b->eq, band, coeffs, enabled);
    return OHS_OK;
}

int ohs_batch_set_stream_eq_band_coeffs(ohs_batch *b, size_t stream, size_t band, const float coeffs[5], int enabled)
{
    if (!b || !coeffs) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    return eq_set_stream_band(b->eq, stream, band, coeffs, enabled);
}

int ohs_batch_update_stream_eq_band(ohs_batch *b, size_t stream, size_t band, float fs, int filter_type, float fc, float q,
                                    float gain_db, int enabled)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    if (stream >= b->conv.S) return fail(OHS_ERR_INVALID_ARG, "stream index out of range");
    if (band >= b->eq.nb) return OHS_OK;
    float c[5];
    int rc = rbj(filter_type, fs, fc, q, gain_db, c);
    if (rc) return rc;
    return eq_set_stream_band(b->eq, stream, band, c, enabled);
}

int ohs_batch_share_eq_table(ohs_batch *b)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    eq_share_table(b->eq);
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
    eq_set_shared_band(b->eq, band, c, enabled);
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
    if (plan < 0 || plan > 2)
        return fail(OHS_ERR_INVALID_ARG, "plan must be 0 (library's choice), 1 (block 512) or 2 (large transforms: hop 1536 / block 2048)");
    b->conv.conv_plan = plan;
    return OHS_OK;
}

int ohs_batch_last_conv_plan(const ohs_batch *b, int *kernel, int *ranges_per_stream)
{
    if (!b || !kernel || !ranges_per_stream) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    *kernel = b->conv.last_kernel;
    *ranges_per_stream = b->conv.last_ranges;
    return OHS_OK;
}

int ohs_batch_conv_plan_counts(ohs_batch *b, uint64_t counts[OHS_CONV_KERNEL_COUNT], int reset)
{
    if (!b || !counts) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    for (int k = 0; k < OHS_CONV_KERNEL_COUNT; ++k) counts[k] = b->conv.kernel_calls[k];
    if (reset)
        for (int k = 0; k < OHS_CONV_KERNEL_COUNT; ++k) b->conv.kernel_calls[k] = 0;
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
    c.lb_lazy = false; c.lb_valid = 0;
    c.pt_active = false;        // (pending tails belong to the frames in front of the reset)
    if (c.d_xhist) {            // (the block-2048 plan's state: zeros are what every path may see of the past)
        HIP_TRY(hipMemsetAsync(c.d_xhist, 0, c.S * 2 * (size_t)(2 * c.xh_len) * sizeof(float), b->st));
        c.xh_valid = c.xh_len;
        c.xh_head = 0;
    }
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
    // (a timing-enabled event: ev_inflight feeds ev_pool, whose events the profiling spans time with)
    if (hipEventCreate(&ev) == hipSuccess) {
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

    const bool eq_active = b->eq_enable && eq_any_enabled(b->eq);

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
    // Fail closed (ohs_batch_reset's contract): a copy, an event or a wait that failed with chunks already processed leaves
    // the per-stream state advanced for some chunks only -- exactly like a failure inside ohs_batch_process, which has
    // marked the handle itself.  (Argument errors were found before the loop: nothing was queued.)
    if (rc && rc != OHS_ERR_INVALID_ARG && !b->failed) {
        const std::string why = g_err;
        b->failed = true;
        b->fail_msg = why;
        return fail(rc, why);
    }
    if (rc) return rc;
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        b->failed = true;
        b->fail_msg = "host pipeline: stream sync failed";
        return fail(OHS_ERR_HIP, b->fail_msg);
    }
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
    HIP_TRY(hipSetDevice(b->device));
    if (b->profiling && !b->eq.d_stamps) {
        HIP_TRY(hipMalloc(&b->eq.d_stamps, 4 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(b->eq.d_stamps, 0, 4 * sizeof(unsigned long long)));
    } else if (!b->profiling && b->eq.d_stamps) {
        DeviceWideSection dws;
        HIP_TRY(hipDeviceSynchronize());        // (a launch in flight may still write them)
        hipFree(b->eq.d_stamps);
        b->eq.d_stamps = nullptr;
    }
    return OHS_OK;
}

int ohs_batch_profile_eq_clock(ohs_batch *b, double *shader_ghz, double *wave_us)
{
    if (!b || !shader_ghz || !wave_us) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    *shader_ghz = 0.0; *wave_us = 0.0;
    if (!b->eq.d_stamps) return fail(OHS_ERR_INVALID_ARG, "profiling is off (ohs_batch_set_profiling)");
    HIP_TRY(hipSetDevice(b->device));
    unsigned long long s[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(s, b->eq.d_stamps, sizeof(s), hipMemcpyDeviceToHost));     // (synchronous: waits for the device)
    if (s[1] <= s[0] || s[3] <= s[2]) return fail(OHS_ERR_INVALID_ARG, "no ring-form EQ launch has run with profiling on");
    *wave_us = (double)(s[1] - s[0]) / 100.0;                                   // s_memrealtime: 100 MHz
    *shader_ghz = (double)(s[3] - s[2]) / ((double)(s[1] - s[0]) * 10.0);       // shader clocks per 10 ns
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
    hipError_t bad = hipSuccess;
    for (auto &sp : b->spans) {
        cntk[sp.kind]++;
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(sp.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, sp.a, sp.b);
        if (e != hipSuccess) {
            // an event that cannot be timed: it does not go back to the pool, and the spans are dropped all the same --
            // one bad span must not make every later read fail
            if (bad == hipSuccess) bad = e;
            hipEventDestroy(sp.a);
            hipEventDestroy(sp.b);
            continue;
        }
        t[sp.kind] += ms;
        b->ev_pool.push_back(sp.a);
        b->ev_pool.push_back(sp.b);
    }
    b->spans.clear();
    if (bad != hipSuccess) {
        b->prof_calls = 0;
        return fail(OHS_ERR_HIP, std::string("profile_read: a span's events could not be timed (") + hipGetErrorString(bad) +
                                     "); the spans recorded so far were dropped");
    }
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
    //   P = 1, block 512: audio in + out; per stream and chunk boundary the input block in front of it once more (4 KiB;
    //                   where the pre-pass computes the boundary tails, K not in {2, 4, 8, 16}: the tail written and read
    //                   as well) and the four-overlap state (8 KiB in + 8 KiB out)
    //   P > 1 (time-parallel): audio in, ring write, ring read once, W write, W read, audio out
    const uint64_t eq_sb = 4096u + 4096u + 520u;
    *eq_bytes = b->eq_enable ? nbk * S * eq_sb : 0;
    const bool os_plan = P == 1 && b->conv.lazy_ok &&
                         (b->conv.conv_plan == 2 ||
                          (b->conv.conv_plan == 0 && conv_plan_auto_is_os((size_t)S, (long long)nbk, b->eq_enable != 0)));   // (EQ on: in place)
    if (os_plan) {
        // hop-1536 plan: audio in + out; per stream the 512 frames in front of every hop range but the first once more
        // (4 KiB each), the merged overlap in + out and the last input block (3 x 4 KiB)
        const uint64_t K = (uint64_t)conv_os_chunks(b->ctx, (size_t)S, (long long)nbk, b->eq_enable != 0);
        *conv_bytes = nbk * S * 8192u + S * ((K - 1) * 4096u + 3u * 4096u);
    } else if (P == 1) {
        const uint64_t K = (uint64_t)conv_p1_chunks(b->ctx, (size_t)S, (long long)nbk, 0);
        const bool own_tails = conv_p1_waves_per_cu() == 16 && (K == 2 || K == 4 || K == 8 || K == 16);
        // state: per-path overlaps 8 KiB in + 8 KiB out, or (lazy, kernels.h) merged overlap 4 KiB in + 4 KiB out + the
        // last block's input copy 4 KiB
        const uint64_t state = b->conv.lazy_ok ? 3u * 4096u : 2u * 8192u;
        *conv_bytes = nbk * S * 8192u + S * ((K - 1) * (own_tails ? 1u : 3u) * 4096u + state);
    } else if (b->conv.d_xhist && P >= lb_min_p() &&
               (b->conv.conv_plan == 2 || (b->conv.conv_plan == 0 && conv_plan_auto_is_lb((size_t)S, (long long)nbk, P)))) {
        if (!b->eq_enable && conv_plan_auto_is_xb((size_t)S, (long long)nbk, P)) {
            // block-8192 kernel (EQ off: taken to be out of place, as the hop-1536 rule above does): audio in + out; per run of
            // blocks the window's first half once more (64 KiB), with two partitions the carry's window as well; the history append
            const int P2x = (P + 15) / 16;
            const uint64_t n_blk = (nbk * 512u + 8191u) / 8192u;
            const uint64_t run = (uint64_t)conv_xb_run_for((int)S, (int)n_blk, P2x, b->ctx->num_cus);
            *conv_bytes = nbk * S * 8192u + S * ((n_blk + run - 1) / run) * (uint64_t)P2x * 65536u + 2u * S * 2u * (uint64_t)b->conv.xh_len * 4u;
            return OHS_OK;
        }
        // block-2048 plan: audio in, ring write (16 B per frame), ring read once, audio out -- the product never leaves the
        // chip; plus the input history copied once per segment (read + write)
        *conv_bytes = nbk * S * (4096u + 8192u + 8192u + 4096u) + 2u * S * 2u * (uint64_t)b->conv.xh_len * 4u;
    } else {
        *conv_bytes = nbk * S * (4096u + 8192u + 8192u + 8192u + 8192u + 4096u);
    }
    return OHS_OK;
}


int ohs_batch_set_speakers(ohs_batch *b, const ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r,
                           float radius_m, float fs, unsigned *changed_mask)
{
    if (!b) return fail(OHS_ERR_INVALID_ARG, "batch is NULL");
    return set_speakers_impl(b->conv, sofa, az_l, el_l, az_r, el_r, radius_m, fs, changed_mask,
                             [&](int p, const float *ir, size_t n) { return ohs_batch_set_ir(b, p, ir, n); });
}

}  // extern "C"
