/* host_node_batch.c -- a compiled host on the node-level batch API of include/ohs_hip.h: the offline many-stream job
 * of BASELINE.json's north_star (independent stereo streams sharded over the GPUs of one node, the shared HRIR set +
 * EQ table broadcast once over RCCL) driven from C, the way the reference's Rust host would drive it through the
 * extern "C" block of INTEGRATION.md section 7.  tests/test_gpu_c_host.py builds it with gcc and checks its output
 * against the oracle.
 *
 *   gcc -O2 -Iinclude -o host_node_batch examples/host_node_batch.c -Lopen_headstage_amd -lohs_hip -lm
 *   ./host_node_batch n_devices streams irs.f32 taps in.f32 out.f32 frames
 *     irs.f32 : 4 x taps f32 (Lsl, Lsr, Rsl, Rsr);  in/out.f32 : planar [streams][2][frames], frames % 512 == 0
 *
 * Built with -DDEVICE_RESIDENT (plus -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include ... -L/opt/rocm/lib -lamdhip64) the
 * host owns the device memory itself -- one buffer per device slot holding that slot's shard -- and drives the node's
 * data path, ohs_node_batch_process / _sync with the per-slot device timer, instead of the PCIe-fed _process_host: what
 * a production host with audio already resident in HBM calls.
 */
#include "ohs_hip.h"
#ifdef DEVICE_RESIDENT
#include <hip/hip_runtime_api.h>
#define HIPCHECK(call)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } \
    } while (0)
#endif

#include <stdio.h>
#include <stdlib.h>

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != OHS_OK) {                                                               \
            fprintf(stderr, "%s -> %s: %s\n", #call, ohs_status_string(rc_), ohs_last_error()); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static float *read_f32(const char *path, size_t n)
{
    FILE *f = fopen(path, "rb");
    float *p = (float *)malloc(n * sizeof(float));
    if (!f || !p || fread(p, sizeof(float), n, f) != n) { fprintf(stderr, "cannot read %zu floats from %s\n", n, path); exit(2); }
    fclose(f);
    return p;
}

int main(int argc, char **argv)
{
    if (argc != 8) { fprintf(stderr, "usage: %s n_devices streams irs.f32 taps in.f32 out.f32 frames\n", argv[0]); return 2; }
    const int n_devices = atoi(argv[1]);
    const size_t streams = (size_t)atol(argv[2]), taps = (size_t)atol(argv[4]), frames = (size_t)atol(argv[7]);
    if (frames % OHS_BLOCK_SIZE) { fprintf(stderr, "frames must be a multiple of %d\n", OHS_BLOCK_SIZE); return 2; }
    float *irs = read_f32(argv[3], 4 * taps);
    float *x = read_f32(argv[5], streams * 2 * frames);
    float *y = (float *)malloc(streams * 2 * frames * sizeof(float));
    if (!y) return 2;

    /* the bench's 10-band table, coefficients by the library's convenience routine */
    enum { NB = 10 };
    static const int types[NB] = {OHS_FILTER_LOWSHELF, OHS_FILTER_PEAK, OHS_FILTER_PEAK, OHS_FILTER_PEAK, OHS_FILTER_PEAK,
                                  OHS_FILTER_PEAK, OHS_FILTER_PEAK, OHS_FILTER_PEAK, OHS_FILTER_PEAK, OHS_FILTER_HIGHSHELF};
    static const float fc[NB] = {105.f, 63.f, 125.f, 250.f, 500.f, 1000.f, 2000.f, 4000.f, 8000.f, 10000.f};
    static const float q[NB] = {0.70f, 1.41f, 1.41f, 1.41f, 1.41f, 1.41f, 1.41f, 1.41f, 1.41f, 0.70f};
    static const float db[NB] = {4.f, -3.f, 2.f, -2.5f, 1.5f, -1.f, 3.f, -4.f, 2.f, -3.f};
    float coeffs[NB * 5];
    int enabled[NB];
    for (int i = 0; i < NB; ++i) {
        CHECK(ohs_biquad_coeffs(types[i], 48000.f, fc[i], q[i], db[i], coeffs + 5 * i));
        enabled[i] = 1;
    }

    ohs_node_batch *nb = NULL;
    CHECK(ohs_node_batch_create(n_devices, NULL, streams, NB, &nb));
    int comm = 0, loaded = 0;
    CHECK(ohs_node_batch_rccl_info(nb, &comm, &loaded));
    for (int s = 0; s < n_devices; ++s) {
        int dev; size_t first, n;
        CHECK(ohs_node_batch_shard(nb, s, &dev, &first, &n));
        fprintf(stderr, "slot %d: device %d, streams [%zu, %zu)\n", s, dev, first, first + n);
    }
    fprintf(stderr, "RCCL communicator of %d rank(s)\n", comm);
    const float *ir_ptrs[4] = {irs, irs + taps, irs + 2 * taps, irs + 3 * taps};
    const size_t lens[4] = {taps, taps, taps, taps};
    CHECK(ohs_node_batch_set_tables(nb, ir_ptrs, lens, coeffs, enabled));     /* ONE broadcast */
    CHECK(ohs_node_batch_set_eq_enabled(nb, 1));
    CHECK(ohs_node_batch_set_gain(nb, 0.8912509f));
#ifdef DEVICE_RESIDENT
    {
        enum { MAX_SLOTS = 64 };
        float *d_buf[MAX_SLOTS];
        const float *d_in[MAX_SLOTS];
        float ms[MAX_SLOTS];
        if (n_devices > MAX_SLOTS) return 2;
        for (int s = 0; s < n_devices; ++s) {           /* each slot's shard onto its own device, in place */
            int dev; size_t first, n;
            CHECK(ohs_node_batch_shard(nb, s, &dev, &first, &n));
            HIPCHECK(hipSetDevice(dev));
            HIPCHECK(hipMalloc((void **)&d_buf[s], n * 2 * frames * sizeof(float)));
            HIPCHECK(hipMemcpy(d_buf[s], x + first * 2 * frames, n * 2 * frames * sizeof(float), hipMemcpyHostToDevice));
            d_in[s] = d_buf[s];
        }
        CHECK(ohs_node_batch_timer_begin(nb));
        CHECK(ohs_node_batch_process(nb, d_in, d_buf, frames / OHS_BLOCK_SIZE, 2 * frames, frames));   /* queued on every device */
        CHECK(ohs_node_batch_timer_end(nb, ms));        /* waits for every device */
        for (int s = 0; s < n_devices; ++s) {
            int dev; size_t first, n;
            CHECK(ohs_node_batch_shard(nb, s, &dev, &first, &n));
            HIPCHECK(hipSetDevice(dev));
            HIPCHECK(hipMemcpy(y + first * 2 * frames, d_buf[s], n * 2 * frames * sizeof(float), hipMemcpyDeviceToHost));
            HIPCHECK(hipFree(d_buf[s]));
            fprintf(stderr, "slot %d: %.3f ms on the device\n", s, ms[s]);
        }
    }
#else
    CHECK(ohs_node_batch_process_host(nb, x, y, frames / OHS_BLOCK_SIZE, 2 * frames, frames, 0));
#endif
    ohs_node_batch_destroy(nb);

    FILE *f = fopen(argv[6], "wb");
    if (!f || fwrite(y, sizeof(float), streams * 2 * frames, f) != streams * 2 * frames) { fprintf(stderr, "cannot write %s\n", argv[6]); return 2; }
    fclose(f);
    free(irs); free(x); free(y);
    return 0;
}
