// api_eq.hip -- the EQ table / state of a handle (EqState), ohs_eq_* (StereoParametricEQ, parametric_eq.rs:125-209)
// and ohs_biquad_* (BiquadFilter as a type of its own, parametric_eq.rs:46-123).
#include "api_internal.h"

using namespace ohs;
using namespace ohs_api;
using ohs_host::rbj;

namespace ohs_api {


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

void eq_free(EqState &e)
{
    if (e.d_state) hipFree(e.d_state);
    if (e.d_stabs) hipFree(e.d_stabs);
    if (e.d_stamps) hipFree(e.d_stamps);
    e.d_state = nullptr; e.d_stabs = nullptr; e.stabs_passes = 0; e.d_stamps = nullptr;
}

bool eq_any_enabled(const EqState &e)
{
    if (e.per_stream) {
        for (unsigned char v : e.s_en)
            if (v) return true;
        return false;
    }
    for (size_t i = 0; i < e.nb; ++i)
        if (e.en[i]) return true;
    return false;
}

void eq_set_shared_band(EqState &e, size_t band, const float coeffs[5], int enabled)
{
    std::memcpy(&e.coeffs[5 * band], coeffs, 5 * sizeof(float));
    e.en[band] = enabled != 0;
    if (e.per_stream) {         // the shared call sets the band of EVERY stream
        const size_t S = e.chains / 2;
        for (size_t s = 0; s < S; ++s) {
            std::memcpy(&e.s_coeffs[(s * e.nb + band) * 5], coeffs, 5 * sizeof(float));
            e.s_en[s * e.nb + band] = enabled != 0;
        }
        e.stabs_dirty = true;
    }
}

int eq_set_stream_band(EqState &e, size_t stream, size_t band, const float coeffs[5], int enabled)
{
    const size_t S = e.chains / 2;
    if (stream >= S) return fail(OHS_ERR_INVALID_ARG, "stream index out of range");
    if (band >= e.nb) return OHS_OK;            // parametric_eq.rs:144-164 ignores a band index past the last one
    if (!e.per_stream) {                        // every stream starts from the shared table
        e.s_coeffs.resize(S * e.nb * 5);
        e.s_en.resize(S * e.nb);
        for (size_t s = 0; s < S; ++s) {
            std::memcpy(&e.s_coeffs[s * e.nb * 5], e.coeffs.data(), e.nb * 5 * sizeof(float));
            for (size_t b = 0; b < e.nb; ++b) e.s_en[s * e.nb + b] = e.en[b] != 0;
        }
        e.per_stream = true;
    }
    std::memcpy(&e.s_coeffs[(stream * e.nb + band) * 5], coeffs, 5 * sizeof(float));
    e.s_en[stream * e.nb + band] = enabled != 0;
    e.stabs_dirty = true;
    return OHS_OK;
}

void eq_share_table(EqState &e)
{
    e.per_stream = false;
    e.s_coeffs.clear(); e.s_coeffs.shrink_to_fit();
    e.s_en.clear(); e.s_en.shrink_to_fit();
    e.stabs_dirty = true;
}

// the cascade of ONE table (coeffs [nb][5], en [nb]) over n frames of `chains` chains: enabled bands only, in cascade order,
// 16 per pass; a disabled band is the identity and keeps its state (parametric_eq.rs:118-120), so it is simply not given
// a lane.  ev_start / ev_stop (optional): recorded at the start of the first and the completion of the last pass
static int eq_launch_table(EqState &e, const float *coeffs, const unsigned char *en8, const int *en32, size_t chains,
                           float *d_state, const float *in, float *out, long long ss, long long cs, long long n,
                           hipStream_t st, bool *did_anything, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    bool did = false;
    const float *src = in;
    std::vector<size_t> on;
    for (size_t b = 0; b < e.nb; ++b)
        if (en8 ? en8[b] != 0 : en32[b] != 0) on.push_back(b);
    for (size_t p0 = 0; p0 < on.size(); p0 += 16) {
        EqPassTable t;
        std::memset(&t, 0, sizeof(t));
        const int nbp = (int)std::min<size_t>(16, on.size() - p0);
        for (int j = 0; j < 16; ++j) {
            const size_t b = on[p0 + (size_t)std::min(j, nbp - 1)];      // unused lanes shadow the last band's slot (never stored)
            t.slot[j] = (int)b;
            if (j < nbp) {
                t.b0[j] = coeffs[5 * b + 0]; t.b1[j] = coeffs[5 * b + 1]; t.b2[j] = coeffs[5 * b + 2];
                t.a1[j] = coeffs[5 * b + 3]; t.a2[j] = coeffs[5 * b + 4];
            }
        }
        int xcd_lo = e.xcd_lo, xcd_n = e.xcd_n;
        if (tuning().eq_xcd_n > 0) { xcd_lo = tuning().eq_xcd_lo; xcd_n = tuning().eq_xcd_n; }
        const bool first = p0 == 0, last = p0 + 16 >= on.size();
        hipError_t err = launch_eq_pass(src, out, ss, cs, n, (int)chains, t, nbp, d_state, st, e.exact_specials, e.fp_mode,
                                        xcd_lo, xcd_n, first ? ev_start : nullptr, last ? ev_stop : nullptr, e.d_stamps);
        if (err != hipSuccess) return fail(OHS_ERR_HIP, std::string("eq launch: ") + hipGetErrorString(err));
        src = out;
        did = true;
    }
    if (did_anything) *did_anything = did;
    return OHS_OK;
}

// per-stream tables: compact every stream's enabled bands into EqStreamTables of 12 and upload them: pass k of stream s, at
// d_stabs[k * S + s], holds that stream's enabled bands 12 k .. 12 k + 11 (none: nb = 0, the pass hands the samples on).  Not on
// the audio path: it waits for what the stream has queued (a launch in flight may still be reading the old tables) and copies
// synchronously.
static int eq_upload_stream_tables(EqState &e, hipStream_t st)
{
    const size_t S = e.chains / 2;
    size_t mx = 0;
    for (size_t s = 0; s < S; ++s) {
        size_t count = 0;
        for (size_t b = 0; b < e.nb; ++b) count += e.s_en[s * e.nb + b] ? 1 : 0;
        mx = std::max(mx, count);
    }
    const size_t passes = std::max<size_t>(1, (mx + 11) / 12);
    std::vector<EqStreamTable> tabs(passes * S);
    std::memset(tabs.data(), 0, tabs.size() * sizeof(EqStreamTable));
    for (size_t s = 0; s < S; ++s) {
        size_t count = 0;
        for (size_t b = 0; b < e.nb; ++b) {
            if (!e.s_en[s * e.nb + b]) continue;
            EqStreamTable &t = tabs[(count / 12) * S + s];
            const int k = (int)(count % 12);
            const float *c = &e.s_coeffs[(s * e.nb + b) * 5];
            t.b0[k] = c[0]; t.b1[k] = c[1]; t.b2[k] = c[2]; t.a1[k] = c[3]; t.a2[k] = c[4];
            t.slot[k] = (int)b;
            t.nb = k + 1;
            ++count;
        }
        for (size_t pk = 0; pk < passes; ++pk) {        // unused entries repeat the pass's last band's slot (never stored)
            EqStreamTable &t = tabs[pk * S + s];
            for (int j = t.nb; j < 12; ++j) t.slot[j] = t.nb ? t.slot[t.nb - 1] : 0;
        }
    }
    e.max_enabled = mx;
    DeviceWideSection dws;
    HIP_TRY(hipStreamSynchronize(st));
    if (!e.d_stabs || e.stabs_passes < passes) {
        if (e.d_stabs) hipFree(e.d_stabs);
        e.d_stabs = nullptr; e.stabs_passes = 0;
        HIP_TRY(hipMalloc(&e.d_stabs, passes * S * sizeof(EqStreamTable)));
        e.stabs_passes = passes;
    }
    HIP_TRY(hipMemcpy(e.d_stabs, tabs.data(), passes * S * sizeof(EqStreamTable), hipMemcpyHostToDevice));
    e.stabs_dirty = false;
    return OHS_OK;
}

// run the cascade over n frames of `chains` chains, in place allowed
int eq_launch(EqState &e, const float *in, float *out, long long ss, long long cs, long long n,
              hipStream_t st, bool *did_anything, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    bool did = false;
    int rc = OHS_OK;
    if (!e.per_stream) {
        rc = eq_launch_table(e, e.coeffs.data(), nullptr, e.en.data(), e.chains, e.d_state, in, out, ss, cs, n, st, &did,
                             ev_start, ev_stop);
        if (rc) return rc;
    } else {
        if (e.stabs_dirty) {
            rc = eq_upload_stream_tables(e, st);
            if (rc) return rc;
        }
        if (e.max_enabled == 0) {
            // no stream has an enabled band: identity
        } else if (!e.exact_specials && !tuning().eq_conveyor && eq_ring_addressable(ss, cs, n)) {
            // one launch per 12 bands for all streams: every row reads its own stream's table when its wave starts (a stream with
            // fewer bands than the pass's first hands its samples on -- exact except for -0.0, as the ring form's spare lanes are)
            int xcd_lo = e.xcd_lo, xcd_n = e.xcd_n;
            if (tuning().eq_xcd_n > 0) { xcd_lo = tuning().eq_xcd_lo; xcd_n = tuning().eq_xcd_n; }
            const size_t S = e.chains / 2, passes = (e.max_enabled + 11) / 12;
            for (size_t pk = 0; pk < passes; ++pk) {
                hipError_t err = launch_eq_ring_streams(pk ? out : in, out, ss, cs, n, (int)e.chains, e.d_stabs + pk * S, e.d_state, st,
                                                        e.fp_mode, xcd_lo, xcd_n, pk == 0 ? ev_start : nullptr,
                                                        pk + 1 == passes ? ev_stop : nullptr);
                if (err != hipSuccess) return fail(OHS_ERR_HIP, std::string("eq launch (per-stream tables): ") + hipGetErrorString(err));
            }
            did = true;
        } else {
            // The exact-specials mode, strides beyond the ring form's reach: the conveyor kernel takes its table as a kernel
            // argument, so every stream is a launch sequence of its own (correct for any table; an offline job that needs this at
            // scale groups its streams by table and uses one batch per group).
            if (ev_start) HIP_TRY(hipEventRecord(ev_start, st));
            const size_t S = e.chains / 2;
            for (size_t s = 0; s < S; ++s) {
                bool d1 = false;
                const float *si = in + (long long)s * ss;
                float *so = out + (long long)s * ss;
                rc = eq_launch_table(e, &e.s_coeffs[s * e.nb * 5], &e.s_en[s * e.nb], nullptr, 2,
                                     e.d_state + s * 2 * (size_t)kEqStateSlots * 2, si, so, ss, cs, n, st, &d1, nullptr, nullptr);
                if (rc) return rc;
                if (!d1 && so != si) {      // this stream has no enabled band: its frames still have to arrive in `out`
                    HIP_TRY(hipMemcpy2DAsync(so, (size_t)cs * sizeof(float), si, (size_t)cs * sizeof(float), (size_t)n * sizeof(float),
                                             2, hipMemcpyDeviceToDevice, st));
                }
                did = did || d1 || so != si;
            }
            if (ev_stop) HIP_TRY(hipEventRecord(ev_stop, st));
            if (did_anything) *did_anything = did;
            return OHS_OK;
        }
    }
    if (!did) {     // no enabled band: the events mark this point of the stream
        if (ev_start) HIP_TRY(hipEventRecord(ev_start, st));
        if (ev_stop) HIP_TRY(hipEventRecord(ev_stop, st));
    }
    if (did_anything) *did_anything = did;
    return OHS_OK;
}

}  // namespace ohs_api

extern "C" {

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
    eq_free(q->eq);
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
    eq_free(f->eq);
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

}  // extern "C"
