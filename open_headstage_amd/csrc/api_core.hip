// api_core.hip -- the C ABI's library-wide part: error state, per-device context (twiddle tables), version / status
// strings, device enumeration, the deployment options of the resident kernels, experiments-build debug entry points.
//
// Host-side orchestration only, here and in the other api_*.hip files: device memory, FIFOs, per-path bookkeeping and
// kernel launches.  Every sample of audio is computed by the HIP kernels in conv_kernels.hip / conv_os_kernels.hip /
// conv_mac_kernels.hip / eq_kernels.hip; there is no CPU compute path and every create call fails with
// OHS_ERR_NO_DEVICE when no GPU is usable.
#include "api_internal.h"

using namespace ohs;
using namespace ohs_api;

namespace ohs_api {

thread_local std::string g_err;
#ifdef OHS_EXPERIMENTS
std::atomic<int> g_inject_batch_failure{0};
#endif

int fail(int status, const std::string &msg)
{
    g_err = msg;
    return status;
}

namespace {
std::mutex g_ctx_mu;
std::vector<DeviceCtx *> g_ctx;
}  // namespace

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
    {
        std::vector<float2> t4(3 * 1024);
        conv_lb_build_twiddles(t4.data());
        if (hipMalloc(&c->d_tw4096, t4.size() * sizeof(float2)) != hipSuccess ||
            hipMemcpy(c->d_tw4096, t4.data(), t4.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
            hipFree(c->d_tw);
            delete c;
            return fail(OHS_ERR_HIP, "w4096 table upload failed");
        }
    }
    {
        std::vector<float2> t16(15 * 1024);
        conv_xb_build_twiddles(t16.data());
        if (hipMalloc(&c->d_tw16384, t16.size() * sizeof(float2)) != hipSuccess ||
            hipMemcpy(c->d_tw16384, t16.data(), t16.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
            hipFree(c->d_tw);
            delete c;
            return fail(OHS_ERR_HIP, "w16384 table upload failed");
        }
    }
    g_ctx.push_back(c);
    *out = c;
    return OHS_OK;
}

RtRegistry g_rt_reg[kMaxDevices];

}  // namespace ohs_api

// =======================================================================================
// C ABI
// =======================================================================================
extern "C" {

// shared with sofa_reader.cpp (not part of the public header)
void ohsint_set_error(const char *msg) { g_err = msg ? msg : ""; }

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
    rt_max_resident_store(max_resident_per_device);     // < 0: the default, a quarter of the device's CUs
    if (idle_polls) rt_idle_polls_store(idle_polls);
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

}  // extern "C"
