/* host_chain.c -- a compiled (non-Python) host on the C ABI of include/ohs_hip.h, doing what
 * Open-Headstage's Plugin::process does per audio callback (src/lib.rs:1169-1207): refresh the EQ
 * coefficients, EQ in place, 4-path HRIR convolution, output gain -- here over a WAV-less stream of
 * 1024-frame host blocks read from / written to raw f32 files.  It is the C twin of the Rust shim in
 * INTEGRATION.md; tests/test_gpu_c_host.py builds it with gcc and checks its output against the oracle.
 *
 *   gcc -O2 -Iinclude -o host_chain examples/host_chain.c -Lopen_headstage_amd -lohs_hip -lm
 *   ./host_chain irs.f32 taps in.f32 out.f32 frames
 *     irs.f32 : 4 x taps f32 (Lsl, Lsr, Rsl, Rsr);  in/out.f32 : planar [2][frames]
 */
#include "ohs_hip.h"

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
    if (argc != 6) { fprintf(stderr, "usage: %s irs.f32 taps in.f32 out.f32 frames\n", argv[0]); return 2; }
    const size_t taps = (size_t)atol(argv[2]), frames = (size_t)atol(argv[5]);
    float *irs = read_f32(argv[1], 4 * taps);
    float *x = read_f32(argv[3], 2 * frames);

    ohs_engine *eng = NULL;
    ohs_eq *eq = NULL;
    CHECK(ohs_engine_create(0, &eng));
    CHECK(ohs_eq_create(0, 10, 48000.0f, &eq));
    for (int p = 0; p < 4; ++p) CHECK(ohs_engine_set_ir(eng, p, irs + (size_t)p * taps, taps));

    /* the 10-band table of SURVEY section 8d (band 0 low shelf, 1..8 peaks, 9 high shelf) */
    static const float fc[10] = {105.f, 63.f, 125.f, 250.f, 500.f, 1000.f, 2000.f, 4000.f, 8000.f, 10000.f};
    static const float q[10] = {0.70f, 1.41f, 1.41f, 1.41f, 1.41f, 1.41f, 1.41f, 1.41f, 1.41f, 0.70f};
    static const float g[10] = {4.f, -3.f, 2.f, -2.5f, 1.5f, -1.f, 3.f, -4.f, 2.f, -3.f};
    const float gain = 0.8912509f;                    /* -1 dB */
    const size_t host_block = 1024;
    for (size_t pos = 0; pos < frames; pos += host_block) {
        const size_t n = frames - pos < host_block ? frames - pos : host_block;
        /* lib.rs:1180-1193: coefficients are refreshed once per host block from the (smoothed) parameters */
        for (int b = 0; b < 10; ++b) {
            const int type = b == 0 ? OHS_FILTER_LOWSHELF : (b == 9 ? OHS_FILTER_HIGHSHELF : OHS_FILTER_PEAK);
            CHECK(ohs_eq_update_band(eq, (size_t)b, 48000.0f, type, fc[b], q[b], g[b], 1));
        }
        /* lib.rs:1194-1207: EQ in place -> convolution -> gain, one call, in place on the host buffers */
        CHECK(ohs_chain_process(eng, eq, /*bypass=*/0, /*eq_enable=*/1, gain, x + pos, x + frames + pos, n));
    }
    FILE *f = fopen(argv[4], "wb");
    if (!f || fwrite(x, sizeof(float), 2 * frames, f) != 2 * frames) { fprintf(stderr, "cannot write %s\n", argv[4]); return 2; }
    fclose(f);
    ohs_eq_destroy(eq);
    ohs_engine_destroy(eng);
    free(irs); free(x);
    printf("processed %zu frames in %zu-frame host blocks\n", frames, host_block);
    return 0;
}
