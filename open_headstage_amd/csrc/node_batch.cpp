// node_batch.cpp -- the offline many-stream batch mode over the GPUs of ONE node, behind the C ABI
// (BASELINE.json north_star: "shards independent stereo streams across the 8 GPUs of one node with RCCL broadcast of
// the shared HRIR set over xGMI").  Host-only C++ on top of the public single-device batch API.
//
// One process, one ohs_batch per device, contiguous stream-id shards, NO data-path collective (all mutable state is
// per stream: convolution.rs:36-42,76-79; parametric_eq.rs:125-129).  The one exchange -- the shared tables -- is a
// real RCCL broadcast: the packed tables are copied to the first device and ncclBroadcast (grouped over the
// communicators ncclCommInitAll returns) carries them to every other device over xGMI; each device then builds its
// spectra from its own copy.  librccl.so is loaded with dlopen when the first node batch is created, so the core
// library has no RCCL dependency and a single-GPU host never loads it.
#include "../../include/ohs_hip.h"
#include "host_internal.h"
#include "tuning.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and enums only; the functions are looked up in the loaded library

#include <dlfcn.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

int fail(int code, const std::string &m) { ohsint_set_error(m.c_str()); return code; }

struct Rccl {
    void *so = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.so) break;
        }
        if (!r.so) { r.error = std::string("librccl.so could not be loaded: ") + dlerror(); return; }
        auto sym = [&](const char *n) -> void * {
            void *p = dlsym(r.so, n);
            if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + n;
            return p;
        };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(sym("ncclCommCuDevice"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

#define NB_HIP(x)                                                                                      \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) return fail(OHS_ERR_HIP, std::string(#x ": ") + hipGetErrorString(e_)); \
    } while (0)
#define NB_NCCL(x)                                                                                         \
    do {                                                                                                   \
        ncclResult_t r_ = (x);                                                                             \
        if (r_ != ncclSuccess) return fail(OHS_ERR_RCCL, std::string(#x ": ") + rccl().GetErrorString(r_)); \
    } while (0)

}  // namespace

// One host thread per device slot, alive as long as the node batch: it has made its device current once and runs
// whatever the calling thread posts (issuing a device's dozen launches from a thread of its own keeps the devices'
// queues filling side by side; the HIP "current device" is per thread, so the caller's own is never touched).
struct DeviceWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, quit = false;
    int rc = OHS_OK;
    std::string err;

    void loop(int device)
    {
        // The thread's current device is what every launch of this slot goes to.  If it cannot be set the slot is dead:
        // its jobs are refused (never run on whatever device the thread happens to have), and the node batch fails closed.
        const hipError_t set = hipSetDevice(device);
        int cur = -1;
        const bool dev_ok = set == hipSuccess && hipGetDevice(&cur) == hipSuccess && cur == device;
        const std::string dev_err = dev_ok ? std::string()
                                           : "worker thread could not make device " + std::to_string(device) + " current (" +
                                                 (set != hipSuccess ? hipGetErrorString(set) : "hipGetDevice disagrees") + ")";
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit) return;
            lk.unlock();
            const int r = dev_ok ? job() : (int)OHS_ERR_HIP;
            std::string e = !dev_ok ? dev_err : (r ? ohs_last_error() : "");      // (thread-local in the core library: carried to the caller)
            lk.lock();
            rc = r; err.swap(e);
            has_job = false;
            cv.notify_all();
        }
    }
    void post(std::function<int()> f)
    {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(f); has_job = true;
        cv.notify_all();
    }
    int wait(std::string *e)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !has_job; });
        if (rc && e) *e = err;
        return rc;
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

struct ohs_node_batch {
    int n = 0;
    size_t streams_total = 0, num_bands = 0;
    std::vector<int> dev;
    std::vector<size_t> first, count;
    std::vector<ohs_batch *> b;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> st;        // one per device, for the broadcasts
    std::vector<hipStream_t> pst;       // one per device: what ohs_node_batch_process queues its device's work on
    std::vector<hipEvent_t> ev0, ev1;   // ohs_node_batch_timer_begin / _end
    std::vector<float *> d_tab;         // broadcast buffer per device
    size_t tab_cap = 0;                 // floats
    std::vector<std::unique_ptr<DeviceWorker>> worker;
    // ohs_node_batch_set_speakers: what it last loaded into each path (any other set_ir / set_tables forgets it)
    std::vector<float> spk_ir[4];
    bool spk_set[4] = {false, false, false, false};
    bool shared_device = false;         // experiments build: slots share a device, no RCCL (Tuning::node_shared_device_rehearsal)
    // fail-closed at node level: a processing call that failed on ANY slot leaves the shards out of step (the others
    // have advanced); a table call that failed half-way leaves the devices with different tables.  Both are refused
    // until the call that repairs them (ohs_node_batch_reset / a complete ohs_node_batch_set_tables) has succeeded.
    bool failed = false, tables_torn = false;
    std::string fail_msg, torn_msg;
};

namespace {

// contiguous, balanced ranges; the first (total % n) shards get one more (open_headstage_amd/sharding.py)
void shard_range(size_t total, int slot, int n, size_t *first, size_t *count)
{
    const size_t q = total / (size_t)n, r = total % (size_t)n;
    *first = (size_t)slot * q + std::min<size_t>((size_t)slot, r);
    *count = q + ((size_t)slot < r ? 1 : 0);
}

int grow_tables(ohs_node_batch *nb, size_t floats)
{
    if (floats <= nb->tab_cap) return OHS_OK;
    size_t cap = nb->tab_cap ? nb->tab_cap : 4096;
    while (cap < floats) cap *= 2;
    nb->tab_cap = 0;        // (a failure half-way leaves buffers of mixed sizes: none of them counts until all are there)
    for (int i = 0; i < nb->n; ++i) {
        NB_HIP(hipSetDevice(nb->dev[(size_t)i]));
        NB_HIP(hipStreamSynchronize(nb->st[(size_t)i]));
        if (nb->d_tab[(size_t)i]) NB_HIP(hipFree(nb->d_tab[(size_t)i]));
        nb->d_tab[(size_t)i] = nullptr;
        NB_HIP(hipMalloc(&nb->d_tab[(size_t)i], cap * sizeof(float)));
    }
    nb->tab_cap = cap;
    return OHS_OK;
}

// host -> first device -> ONE grouped ncclBroadcast -> every device; returns with the data in place everywhere
int broadcast_from_host(ohs_node_batch *nb, const float *host, size_t floats)
{
    if (floats == 0) return OHS_OK;
    int rc = grow_tables(nb, floats);
    if (rc) return rc;
    NB_HIP(hipSetDevice(nb->dev[0]));
    NB_HIP(hipMemcpyAsync(nb->d_tab[0], host, floats * sizeof(float), hipMemcpyHostToDevice, nb->st[0]));
    if (nb->shared_device) {        // rehearsal on one GPU: slot 0's buffer to the other slots by device copies
        NB_HIP(hipStreamSynchronize(nb->st[0]));
        for (int i = 1; i < nb->n; ++i) {
            NB_HIP(hipSetDevice(nb->dev[(size_t)i]));
            NB_HIP(hipMemcpyAsync(nb->d_tab[(size_t)i], nb->d_tab[0], floats * sizeof(float), hipMemcpyDeviceToDevice, nb->st[(size_t)i]));
            NB_HIP(hipStreamSynchronize(nb->st[(size_t)i]));
        }
        return OHS_OK;
    }
    Rccl &R = rccl();
    NB_NCCL(R.GroupStart());
    for (int i = 0; i < nb->n; ++i) {
        const ncclResult_t r = R.Broadcast(nb->d_tab[0], nb->d_tab[(size_t)i], floats, ncclFloat, 0, nb->comm[(size_t)i],
                                           nb->st[(size_t)i]);
        if (r != ncclSuccess) { R.GroupEnd(); return fail(OHS_ERR_RCCL, std::string("ncclBroadcast: ") + R.GetErrorString(r)); }
    }
    NB_NCCL(R.GroupEnd());
    for (int i = 0; i < nb->n; ++i) {
        NB_HIP(hipSetDevice(nb->dev[(size_t)i]));
        NB_HIP(hipStreamSynchronize(nb->st[(size_t)i]));
    }
    return OHS_OK;
}

// the calling thread's current device is the caller's: entry points that visit the devices themselves put it back
struct DeviceRestore {
    int dev = 0;
    bool ok = false;
    DeviceRestore() { ok = hipGetDevice(&dev) == hipSuccess; }
    ~DeviceRestore() { if (ok) hipSetDevice(dev); }
};

// fn(slot) on every device's worker thread at once; the first failure (by slot) is reported with its slot number
int run_on_all(ohs_node_batch *nb, const std::function<int(int)> &fn)
{
    for (int i = 0; i < nb->n; ++i) nb->worker[(size_t)i]->post([&fn, i] { return fn(i); });
    int first_rc = OHS_OK;
    std::string first_err;
    int first_slot = -1;
    for (int i = 0; i < nb->n; ++i) {
        std::string e;
        const int r = nb->worker[(size_t)i]->wait(&e);
        if (r && first_rc == OHS_OK) { first_rc = r; first_err = e; first_slot = i; }
    }
    if (first_rc) return fail(first_rc, "device slot " + std::to_string(first_slot) + ": " + first_err);
    return OHS_OK;
}

int refuse_if_unusable(const ohs_node_batch *nb)
{
    if (nb->failed)
        return fail(OHS_ERR_HIP, "an earlier processing call failed on one device (" + nb->fail_msg + "): the shards' per-stream "
                    "state is half-advanced and out of step; ohs_node_batch_reset first");
    if (nb->tables_torn)
        return fail(OHS_ERR_HIP, "an earlier table call failed half-way (" + nb->torn_msg + "): the devices hold different "
                    "tables; repeat ohs_node_batch_set_tables first");
    return OHS_OK;
}

// (OHS_ERR_INVALID_ARG: every slot's ohs_batch_* call checks its arguments before it queues or stores anything -- the
// same arguments on every slot, so all of them refused and nothing is out of step or torn)
int processing_result(ohs_node_batch *nb, int rc)
{
    if (rc && rc != OHS_ERR_INVALID_ARG) { nb->failed = true; nb->fail_msg = ohs_last_error(); }
    return rc;
}

int table_result(ohs_node_batch *nb, int rc)
{
    if (rc && rc != OHS_ERR_INVALID_ARG) { nb->tables_torn = true; nb->torn_msg = ohs_last_error(); }
    return rc;
}

// what ncclCommInitAll returned is what was asked for: one communicator per device slot, rank i on device dev[i], all
// of size n (a mismatch would make the grouped broadcast write a device other than the one its ohs_batch lives on)
int check_communicators(ohs_node_batch *nb)
{
    Rccl &R = rccl();
    for (int i = 0; i < nb->n; ++i) {
        if (!nb->comm[(size_t)i]) return fail(OHS_ERR_RCCL, "ncclCommInitAll returned no communicator for slot " + std::to_string(i));
        int cnt = -1, cu = -1, rank = -1;
        NB_NCCL(R.CommCount(nb->comm[(size_t)i], &cnt));
        NB_NCCL(R.CommCuDevice(nb->comm[(size_t)i], &cu));
        NB_NCCL(R.CommUserRank(nb->comm[(size_t)i], &rank));
        if (cnt != nb->n || cu != nb->dev[(size_t)i] || rank != i)
            return fail(OHS_ERR_RCCL, "ncclCommInitAll: slot " + std::to_string(i) + " got a communicator of size " + std::to_string(cnt) +
                        " (wanted " + std::to_string(nb->n) + "), rank " + std::to_string(rank) + " (wanted " + std::to_string(i) +
                        ") on device " + std::to_string(cu) + " (wanted " + std::to_string(nb->dev[(size_t)i]) + ")");
    }
    return OHS_OK;
}

}  // namespace

extern "C" {

int ohs_node_batch_create(int n_devices, const int *devices, size_t streams_total, size_t num_bands, ohs_node_batch **out)
{
    if (!out) return fail(OHS_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return fail(OHS_ERR_INVALID_ARG, "n_devices must be 1..64");
    if (streams_total < (size_t)n_devices) return fail(OHS_ERR_INVALID_ARG, "fewer streams than devices: a device would have no stream");
    int have = 0;
    int rc = ohs_device_count(&have);
    if (rc) return fail(rc, std::string(ohs_last_error()) + " -- the library has no CPU fallback");
    std::vector<int> dev((size_t)n_devices);
    bool shared = false;
    for (int i = 0; i < n_devices; ++i) {
        dev[(size_t)i] = devices ? devices[i] : i;
        if (dev[(size_t)i] < 0 || dev[(size_t)i] >= have) return fail(OHS_ERR_INVALID_ARG, "device index out of range");
        for (int k = 0; k < i; ++k)
            if (dev[(size_t)k] == dev[(size_t)i]) {
                if (!ohs::tuning().node_shared_device_rehearsal) return fail(OHS_ERR_INVALID_ARG, "a device is listed twice");
                shared = true;
            }
    }
    Rccl &R = rccl();
    if (!shared && !R.error.empty()) return fail(OHS_ERR_RCCL, R.error);
    ohs_node_batch *nb = new (std::nothrow) ohs_node_batch;
    if (!nb) return fail(OHS_ERR_ALLOC, "out of host memory");
    nb->n = n_devices; nb->streams_total = streams_total; nb->num_bands = num_bands; nb->dev = dev;
    nb->shared_device = shared;
    nb->first.resize((size_t)n_devices); nb->count.resize((size_t)n_devices);
    nb->b.assign((size_t)n_devices, nullptr); nb->comm.assign((size_t)n_devices, nullptr);
    nb->st.assign((size_t)n_devices, nullptr); nb->d_tab.assign((size_t)n_devices, nullptr);
    nb->pst.assign((size_t)n_devices, nullptr);
    nb->ev0.assign((size_t)n_devices, nullptr); nb->ev1.assign((size_t)n_devices, nullptr);
    DeviceRestore restore;
    auto body = [&]() -> int {
        for (int i = 0; i < n_devices; ++i) {
            shard_range(streams_total, i, n_devices, &nb->first[(size_t)i], &nb->count[(size_t)i]);
            int r = ohs_batch_create(dev[(size_t)i], nb->count[(size_t)i], num_bands, &nb->b[(size_t)i]);
            if (r) return r;
            NB_HIP(hipSetDevice(dev[(size_t)i]));
            NB_HIP(hipStreamCreateWithFlags(&nb->st[(size_t)i], hipStreamNonBlocking));
            NB_HIP(hipStreamCreateWithFlags(&nb->pst[(size_t)i], hipStreamNonBlocking));
            NB_HIP(hipEventCreate(&nb->ev0[(size_t)i]));
            NB_HIP(hipEventCreate(&nb->ev1[(size_t)i]));
        }
        if (!shared) {
            NB_NCCL(R.CommInitAll(nb->comm.data(), n_devices, dev.data()));
            if (const int r = check_communicators(nb)) return r;
        }
        for (int i = 0; i < n_devices; ++i) {
            nb->worker.emplace_back(new DeviceWorker);
            DeviceWorker *w = nb->worker.back().get();
            w->th = std::thread([w, d = dev[(size_t)i]] { w->loop(d); });
        }
        return OHS_OK;
    };
    rc = body();
    if (rc) {
        const std::string keep = ohs_last_error();
        ohs_node_batch_destroy(nb);
        return fail(rc, keep);
    }
    *out = nb;
    return OHS_OK;
}

void ohs_node_batch_destroy(ohs_node_batch *nb)
{
    if (!nb) return;
    DeviceRestore restore;
    Rccl &R = rccl();
    for (auto &w : nb->worker) w->stop();
    for (int i = 0; i < nb->n; ++i) {
        hipSetDevice(nb->dev[(size_t)i]);
        if (nb->pst[(size_t)i]) { hipStreamSynchronize(nb->pst[(size_t)i]); hipStreamDestroy(nb->pst[(size_t)i]); }
        if (nb->ev0[(size_t)i]) hipEventDestroy(nb->ev0[(size_t)i]);
        if (nb->ev1[(size_t)i]) hipEventDestroy(nb->ev1[(size_t)i]);
        if (nb->st[(size_t)i]) hipStreamSynchronize(nb->st[(size_t)i]);
        if (nb->comm[(size_t)i] && R.CommDestroy) R.CommDestroy(nb->comm[(size_t)i]);
        if (nb->d_tab[(size_t)i]) hipFree(nb->d_tab[(size_t)i]);
        if (nb->st[(size_t)i]) hipStreamDestroy(nb->st[(size_t)i]);
        if (nb->b[(size_t)i]) ohs_batch_destroy(nb->b[(size_t)i]);
    }
    delete nb;
}

int ohs_node_batch_shard(const ohs_node_batch *nb, int slot, int *device, size_t *first_stream, size_t *n_streams)
{
    if (!nb || slot < 0 || slot >= nb->n) return fail(OHS_ERR_INVALID_ARG, "bad node batch / slot");
    if (device) *device = nb->dev[(size_t)slot];
    if (first_stream) *first_stream = nb->first[(size_t)slot];
    if (n_streams) *n_streams = nb->count[(size_t)slot];
    return OHS_OK;
}

int ohs_node_batch_device_batch(ohs_node_batch *nb, int slot, ohs_batch **out)
{
    if (!nb || !out || slot < 0 || slot >= nb->n) return fail(OHS_ERR_INVALID_ARG, "bad node batch / slot");
    *out = nb->b[(size_t)slot];
    return OHS_OK;
}

int ohs_node_batch_set_tables(ohs_node_batch *nb, const float *const irs[4], const size_t lens[4],
                              const float *eq_coeffs, const int *eq_enabled)
{
    if (!nb || !irs || !lens) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    if ((eq_coeffs == nullptr) != (eq_enabled == nullptr)) return fail(OHS_ERR_INVALID_ARG, "eq_coeffs and eq_enabled go together");
    DeviceRestore restore;
    size_t off[5] = {0, 0, 0, 0, 0};
    for (int p = 0; p < 4; ++p) {
        if (lens[p] && !irs[p]) return fail(OHS_ERR_INVALID_ARG, "ir is NULL");
        if (lens[p] > (size_t)1 << 30) return fail(OHS_ERR_INVALID_ARG, "ir too long");
        off[p + 1] = off[p] + lens[p];
    }
    const size_t eq_floats = eq_coeffs ? nb->num_bands * 6 : 0;         // 5 coefficients + the enable flag per band
    std::vector<float> pack(off[4] + eq_floats);
    for (int p = 0; p < 4; ++p)
        if (lens[p]) std::memcpy(pack.data() + off[p], irs[p], lens[p] * sizeof(float));
    for (size_t i = 0; i < (eq_coeffs ? nb->num_bands : 0); ++i) {
        std::memcpy(pack.data() + off[4] + 6 * i, eq_coeffs + 5 * i, 5 * sizeof(float));
        pack[off[4] + 6 * i + 5] = eq_enabled[i] ? 1.0f : 0.0f;
    }
    for (int p = 0; p < 4; ++p) nb->spk_set[p] = false;
    auto unpack = [&]() -> int {
        int rc = broadcast_from_host(nb, pack.data(), pack.size());
        if (rc) return rc;
        std::vector<float> eq_back(eq_floats);
        for (int i = 0; i < nb->n; ++i) {
            if (ohs::tuning().node_inject_table_failure_slot == i + 1)     // (experiments build only; 0 in the product)
                return fail(OHS_ERR_HIP, "injected table failure (node_inject_table_failure_slot)");
            for (int p = 0; p < 4; ++p) {
                rc = ohsint_batch_set_ir_device(nb->b[(size_t)i], p, nb->d_tab[(size_t)i] + off[p], lens[p]);
                if (rc) return rc;
            }
            if (eq_floats) {
                // the EQ table is host-side state of a handle (kernel arguments): every device's handle gets the bits
                // its OWN device received
                NB_HIP(hipSetDevice(nb->dev[(size_t)i]));
                NB_HIP(hipMemcpy(eq_back.data(), nb->d_tab[(size_t)i] + off[4], eq_floats * sizeof(float), hipMemcpyDeviceToHost));
                for (size_t k = 0; k < nb->num_bands; ++k) {
                    rc = ohs_batch_set_eq_band_coeffs(nb->b[(size_t)i], k, eq_back.data() + 6 * k, eq_back[6 * k + 5] != 0.0f);
                    if (rc) return rc;
                }
            }
        }
        return OHS_OK;
    };
    const int rc = table_result(nb, unpack());
    // a COMPLETE table set (four paths and the EQ) repairs whatever an earlier table call tore; without the EQ part only
    // if it was not an EQ call that tore it -- kept simple: the four paths and the EQ together
    if (rc == OHS_OK && eq_coeffs) nb->tables_torn = false;
    return rc;
}

int ohs_node_batch_set_ir(ohs_node_batch *nb, int path, const float *ir, size_t len)
{
    if (!nb) return fail(OHS_ERR_INVALID_ARG, "node batch is NULL");
    if (path < 0 || path > 3) return fail(OHS_ERR_INVALID_ARG, "path must be 0..3");
    if (len && !ir) return fail(OHS_ERR_INVALID_ARG, "ir is NULL");
    if (len > (size_t)1 << 30) return fail(OHS_ERR_INVALID_ARG, "ir too long");
    DeviceRestore restore;
    nb->spk_set[path] = false;
    auto unpack = [&]() -> int {
        int rc = broadcast_from_host(nb, ir, len);
        if (rc) return rc;
        for (int i = 0; i < nb->n; ++i) {
            rc = ohsint_batch_set_ir_device(nb->b[(size_t)i], path, nb->d_tab[(size_t)i], len);
            if (rc) return rc;
        }
        return OHS_OK;
    };
    return table_result(nb, unpack());
}

// speaker angles -> four impulse responses (speakers.cpp), one broadcast per path whose response changed
int ohs_node_batch_set_speakers(ohs_node_batch *nb, const ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r,
                                float radius_m, float fs, unsigned *changed_mask)
{
    if (changed_mask) *changed_mask = 0;
    if (!nb || !sofa) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    std::vector<float> irs[4];
    int rc = ohs_host::speaker_irs(sofa, az_l, el_l, az_r, el_r, radius_m, fs, irs);
    if (rc) return rc;
    for (int p = 0; p < 4; ++p) {
        if (nb->spk_set[p] && nb->spk_ir[p] == irs[p]) continue;
        rc = ohs_node_batch_set_ir(nb, p, irs[p].data(), irs[p].size());
        if (rc) return rc;
        nb->spk_ir[p].swap(irs[p]);
        nb->spk_set[p] = true;
        if (changed_mask) *changed_mask |= 1u << p;
    }
    return OHS_OK;
}

#define NB_FOR_EACH(call)                                                         \
    if (!nb) return fail(OHS_ERR_INVALID_ARG, "node batch is NULL");              \
    for (int i = 0; i < nb->n; ++i) {                                             \
        ohs_batch *h = nb->b[(size_t)i];                                          \
        const int rc = (call);                                                    \
        if (rc) return rc;                                                        \
    }                                                                             \
    return OHS_OK;

int ohs_node_batch_set_eq_band_coeffs(ohs_node_batch *nb, size_t band, const float coeffs[5], int enabled)
{
    if (!nb) return fail(OHS_ERR_INVALID_ARG, "node batch is NULL");
    if (band >= nb->num_bands) return OHS_OK;       // parametric_eq.rs:144-164 ignores a band index past the last one
    auto each = [&]() -> int { NB_FOR_EACH(ohs_batch_set_eq_band_coeffs(h, band, coeffs, enabled)) };
    return table_result(nb, each());
}
// one stream's own band (every plugin instance of the reference owns its bands, parametric_eq.rs:125-129): host-side state of
// the ONE device handle that holds the stream -- nothing to broadcast, nothing that could tear
int ohs_node_batch_set_stream_eq_band_coeffs(ohs_node_batch *nb, size_t stream, size_t band, const float coeffs[5], int enabled)
{
    if (!nb || !coeffs) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    if (stream >= nb->streams_total) return fail(OHS_ERR_INVALID_ARG, "stream index out of range");
    for (int i = 0; i < nb->n; ++i)
        if (stream < nb->first[(size_t)i] + nb->count[(size_t)i])
            return ohs_batch_set_stream_eq_band_coeffs(nb->b[(size_t)i], stream - nb->first[(size_t)i], band, coeffs, enabled);
    return fail(OHS_ERR_INVALID_ARG, "stream index out of range");
}
int ohs_node_batch_share_eq_table(ohs_node_batch *nb) { NB_FOR_EACH(ohs_batch_share_eq_table(h)) }
int ohs_node_batch_set_eq_enabled(ohs_node_batch *nb, int eq_enable) { NB_FOR_EACH(ohs_batch_set_eq_enabled(h, eq_enable)) }
int ohs_node_batch_set_gain(ohs_node_batch *nb, float gain) { NB_FOR_EACH(ohs_batch_set_gain(h, gain)) }
int ohs_node_batch_set_conv_plan(ohs_node_batch *nb, int plan) { NB_FOR_EACH(ohs_batch_set_conv_plan(h, plan)) }
int ohs_node_batch_reset(ohs_node_batch *nb)
{
    DeviceRestore restore;
    auto each = [&]() -> int { NB_FOR_EACH(ohs_batch_reset(h)) };
    const int rc = each();
    if (rc == OHS_OK) { nb->failed = false; nb->fail_msg.clear(); }     // every shard is back at zeroed state, tables kept
    return rc;
}

int ohs_node_batch_process_host(ohs_node_batch *nb, const float *h_in, float *h_out, size_t n_blocks,
                                size_t stream_stride, size_t channel_stride, size_t chunk_blocks)
{
    if (!nb || !h_in || !h_out) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    if (const int r = refuse_if_unusable(nb)) return r;
    return processing_result(nb, run_on_all(nb, [&](int i) {
        const size_t o = nb->first[(size_t)i] * stream_stride;
        return ohs_batch_process_host(nb->b[(size_t)i], h_in + o, h_out + o, n_blocks, stream_stride, channel_stride,
                                      chunk_blocks);
    }));
}

// Device-resident data path: every device's ohs_batch_process is issued by that device's own host thread on that
// device's own stream; nothing crosses PCIe or xGMI (streams do not interact: convolution.rs:36-42,76-79).
int ohs_node_batch_process(ohs_node_batch *nb, const float *const d_in[], float *const d_out[], size_t n_blocks,
                           size_t stream_stride, size_t channel_stride)
{
    if (!nb || !d_in || !d_out) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    for (int i = 0; i < nb->n; ++i)
        if (!d_in[i] || !d_out[i]) return fail(OHS_ERR_INVALID_ARG, "NULL device pointer for slot " + std::to_string(i));
    if (const int r = refuse_if_unusable(nb)) return r;
    return processing_result(nb, run_on_all(nb, [&](int i) {
        return ohs_batch_process(nb->b[(size_t)i], d_in[i], d_out[i], n_blocks, stream_stride, channel_stride,
                                 nb->pst[(size_t)i]);
    }));
}

int ohs_node_batch_sync(ohs_node_batch *nb)
{
    if (!nb) return fail(OHS_ERR_INVALID_ARG, "node batch is NULL");
    // every device is waited for in any case (nothing may still be running when the caller looks at its buffers); a
    // failed / torn node batch then says so instead of "OK": what the buffers hold is not a result
    const int rc = run_on_all(nb, [&](int i) { return ohs_batch_sync(nb->b[(size_t)i], nb->pst[(size_t)i]); });
    if (const int r = refuse_if_unusable(nb)) return r;
    return rc;
}

int ohs_node_batch_stream(ohs_node_batch *nb, int slot, void **hip_stream)
{
    if (!nb || !hip_stream || slot < 0 || slot >= nb->n) return fail(OHS_ERR_INVALID_ARG, "bad node batch / slot");
    *hip_stream = (void *)nb->pst[(size_t)slot];
    return OHS_OK;
}

int ohs_node_batch_timer_begin(ohs_node_batch *nb)
{
    if (!nb) return fail(OHS_ERR_INVALID_ARG, "node batch is NULL");
    if (const int r = refuse_if_unusable(nb)) return r;
    return run_on_all(nb, [&](int i) {
        NB_HIP(hipEventRecord(nb->ev0[(size_t)i], nb->pst[(size_t)i]));
        return (int)OHS_OK;
    });
}

int ohs_node_batch_timer_end(ohs_node_batch *nb, float *ms_per_slot)
{
    if (!nb || !ms_per_slot) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    if (const int r = refuse_if_unusable(nb)) return r;     // (a time measured over a half-advanced step is not a step time)
    return run_on_all(nb, [&](int i) {
        NB_HIP(hipEventRecord(nb->ev1[(size_t)i], nb->pst[(size_t)i]));
        const int r = ohs_batch_sync(nb->b[(size_t)i], nb->pst[(size_t)i]);
        if (r) return r;
        NB_HIP(hipEventElapsedTime(&ms_per_slot[i], nb->ev0[(size_t)i], nb->ev1[(size_t)i]));
        return (int)OHS_OK;
    });
}

int ohs_node_batch_rccl_info(const ohs_node_batch *nb, int *comm_size, int *rccl_loaded)
{
    if (!nb) return fail(OHS_ERR_INVALID_ARG, "node batch is NULL");
    Rccl &R = rccl();
    if (rccl_loaded) *rccl_loaded = R.so != nullptr && R.error.empty();
    if (comm_size) {
        int n = 0;
        if (!nb->shared_device) NB_NCCL(R.CommCount(nb->comm[0], &n));      // (shared-device rehearsal: no communicator, 0)
        *comm_size = n;
    }
    return OHS_OK;
}

}  // extern "C"
