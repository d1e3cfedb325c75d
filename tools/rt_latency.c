/* Per-call latency of ConvolutionEngine::process_block through the C ABI, as a native audio host sees it
 * (no Python in the loop): launch-per-call path vs the resident real-time kernel, 1024-frame host blocks and a
 * near-empty call (mailbox round trip).  4 x 512-tap HRIRs.
 *   gcc -O2 -std=c99 -Iinclude -o tools/bin/rt_latency tools/rt_latency.c -Lopen_headstage_amd -lohs_hip -lm \
 *       -Wl,-rpath,/root/repo/open_headstage_amd
 *   gpurun -- tools/bin/rt_latency */
#define _POSIX_C_SOURCE 199309L
#include "ohs_hip.h"
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_us(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}
static int cmp(const void *a, const void *b) { const double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y; }

/* "1.23" for b >= a within a second of each other (10 ns ticks), "n/a" for a phase that did not run */
static const char *fmt_phase(char *buf, size_t cap, uint64_t a, uint64_t b)
{
    const int64_t d = (int64_t)(b - a);
    if (a == 0 || b == 0 || d < 0 || d > 100000000) snprintf(buf, cap, "n/a");
    else snprintf(buf, cap, "%.2f", (double)d * 0.01);
    return buf;
}

static double g_gap_us = 0.0;    /* > 0: the host idles that long in front of every call (not timed), as a real-time host does */
static ohs_eq *g_eq = NULL;      /* != NULL: time ohs_chain_process (EQ -> convolution -> gain) instead of process_block */

static int one_call(ohs_engine *e, float *il, float *ir, float *ol, float *orr, size_t n)
{
    if (!g_eq) return ohs_engine_process_block(e, il, ir, ol, orr, n);
    memcpy(ol, il, n * 4); memcpy(orr, ir, n * 4);          /* the chain works in place */
    /* per-callback coefficient refresh, as lib.rs:1180-1193 does it */
    for (size_t b = 0; b < 10; ++b)
        ohs_eq_update_band(g_eq, b, 48000.0f, b == 0 ? OHS_FILTER_LOWSHELF : (b == 9 ? OHS_FILTER_HIGHSHELF : OHS_FILTER_PEAK),
                           60.0f * (float)(1 << b) > 18000.0f ? 18000.0f : 60.0f * (float)(1 << b), 1.2f, (b & 1) ? 2.0f : -2.5f, 1);
    return ohs_chain_process(e, g_eq, 0, 1, 0.9f, ol, orr, n);
}

static void run(ohs_engine *e, size_t n, int calls, const char *what)
{
    float *il = calloc(n, 4), *ir = calloc(n, 4), *ol = calloc(n, 4), *orr = calloc(n, 4);
    double *t = malloc(sizeof(double) * (size_t)calls);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; il[i] = (float)(s >> 8) / 8388608.0f - 1.0f; ir[i] = -il[i]; }
    for (int c = 0; c < 50; ++c) one_call(e, il, ir, ol, orr, n);
    for (int c = 0; c < calls; ++c) {
        if (g_gap_us > 0.0) { const double tw = now_us() + g_gap_us; while (now_us() < tw) { } }
        const double t0 = now_us();
        if (one_call(e, il, ir, ol, orr, n)) { printf("error: %s\n", ohs_last_error()); exit(1); }
        t[c] = now_us() - t0;
    }
    {   /* device-side phases of the last call, if the resident kernel served it (10 ns ticks).  A phase that did not
         * run leaves its stamp behind the next one's (or at zero): differences are taken signed and printed as n/a */
        uint64_t st[16];
        char buf[8][24];
#define PHASE(k, a, b) (fmt_phase(buf[k], sizeof(buf[k]), st[a], st[b]))
        if (ohs_engine_debug_rt_stamps(e, st) == 0 && st[4] > st[0] && st[4] - st[0] < 100000)
            printf("# device phases of the last call [us]: acquire %s, body %s, drain+barrier %s, release %s, total %s; shader clock %.0f MHz\n",
                   PHASE(0, 0, 1), PHASE(1, 1, 2), PHASE(2, 2, 3), PHASE(3, 3, 4), PHASE(4, 0, 4),
                   (double)st[5] / ((double)(int64_t)(st[4] - st[0]) * 0.01));
        if (g_eq && st[7] > st[6] && st[7] - st[6] < 100000)
            printf("# EQ inside the resident kernel [us]: copy-in + table %s, EQ %s\n", PHASE(0, 0, 6), PHASE(1, 6, 7));
        if (ohs_engine_debug_rt_stamps(e, st) == 0 && st[14] > st[8] && st[14] - st[8] < 100000)
            printf("# inside the body [us]: pop %s, load+fwd FFT %s, barrier %s, MAC %s, 2 inverse FFTs %s, tails+barrier+store %s\n",
                   PHASE(0, 8, 9), PHASE(1, 9, 10), PHASE(2, 10, 11), PHASE(3, 11, 12), PHASE(4, 12, 13), PHASE(5, 13, 14));
#undef PHASE
    }
    qsort(t, (size_t)calls, sizeof(double), cmp);
    if (calls >= 100000)
        printf("{\"config\": \"%s\", \"frames\": %zu, \"calls\": %d, \"us_median\": %.1f, \"us_p99\": %.1f, \"us_p999\": %.1f, "
               "\"us_p9999\": %.1f, \"us_max\": %.1f, \"us_min\": %.1f}\n", what, n, calls, t[calls / 2], t[(int)(calls * 0.99)],
               t[(int)(calls * 0.999)], t[(int)(calls * 0.9999)], t[calls - 1], t[0]);
    else
    printf("{\"config\": \"%s\", \"frames\": %zu, \"us_median\": %.1f, \"us_p99\": %.1f, \"us_max\": %.1f, \"us_min\": %.1f}\n", what, n,
           t[calls / 2], t[(int)(calls * 0.99)], t[calls - 1], t[0]);
    free(il); free(ir); free(ol); free(orr); free(t);
}

int main(int argc, char **argv)
{
    ohs_engine *e = NULL;
    /* rt_latency soak N: N calls of the resident 1024-frame configuration only, with the tail of the distribution */
    const int soak = (argc > 2 && !strcmp(argv[1], "soak")) ? atoi(argv[2]) : 0;
    if (ohs_engine_create(0, &e)) { printf("create: %s\n", ohs_last_error()); return 1; }
    float h[512];
    for (int p = 0; p < 4; ++p) {
        for (int i = 0; i < 512; ++i) h[i] = (i == 30 + 15 * (p == 1 || p == 2)) ? 1.0f : 0.3f * expf(-i / 85.0f) * ((i * 7919 % 13) - 6) / 60.0f;
        ohs_engine_set_ir(e, p, h, 512);
    }
    if (argc > 1 && !strcmp(argv[1], "long")) {
        /* rt_latency long: the plugin's 1 024-frame call on ROOM responses (BASELINE config 4's 4 x 16 384 taps, and 4 096):
         * calls back to back -- each waits for what the previous one left to compute ahead -- and 300 us apart (a real-time
         * host's are 21 ms apart) */
        static float hl[16384];
        const int taps_list[2] = {4096, 16384};
        for (int k = 0; k < 2; ++k) {
            const int taps = taps_list[k];
            char name[96];
            for (int p = 0; p < 4; ++p) {
                for (int i = 0; i < taps; ++i)
                    hl[i] = (i == 30 + 15 * (p == 1 || p == 2)) ? 1.0f : 0.3f * expf(-(float)i * 6.0f / (float)taps) * ((i * 7919 % 13) - 6) / (float)taps;
                if (ohs_engine_set_ir(e, p, hl, (size_t)taps)) { printf("set_ir: %s\n", ohs_last_error()); return 1; }
            }
            for (int rt = 0; rt < 2; ++rt) {
                ohs_engine_set_realtime(e, rt);
                for (int paced = 0; paced < 2; ++paced) {
                    g_gap_us = paced ? 300.0 : 0.0;
                    snprintf(name, sizeof(name), "c_host_engine_1024_%dtaps_%s_%s", taps, rt ? "resident_kernel" : "launch_per_call",
                             paced ? "300us_apart" : "back_to_back");
                    run(e, 1024, 1000, name);
                }
            }
            g_gap_us = 0.0;
            ohs_engine_set_realtime(e, 0);
        }
        ohs_engine_destroy(e);
        return 0;
    }
    if (soak > 0) {
        ohs_engine_set_realtime(e, 1);
        run(e, 1024, soak, "c_host_engine_1024_resident_kernel_soak");
        ohs_engine_set_realtime(e, 0);
        ohs_engine_destroy(e);
        return 0;
    }
    run(e, 1024, 2000, "c_host_engine_1024_launch_per_call");
    ohs_engine_set_realtime(e, 1);
    run(e, 1024, 2000, "c_host_engine_1024_resident_kernel");
    run(e, 8, 2000, "c_host_mailbox_round_trip_8_frames");
    run(e, 256, 2000, "c_host_engine_256_resident_kernel");
    ohs_engine_set_realtime(e, 0);
    run(e, 256, 2000, "c_host_engine_256_launch_per_call");
    /* the plugin's whole chain: 10-band EQ (coefficients refreshed per call) -> convolution -> gain */
    if (ohs_eq_create(0, 10, 48000.0f, &g_eq)) { printf("eq create: %s\n", ohs_last_error()); return 1; }
    run(e, 1024, 2000, "c_host_chain_eq_conv_gain_1024_launch_per_call");
    ohs_engine_set_realtime(e, 1);
    run(e, 1024, 2000, "c_host_chain_eq_conv_gain_1024_resident_kernel");
    ohs_engine_set_realtime(e, 0);
    ohs_eq_destroy(g_eq);
    ohs_engine_destroy(e);
    return 0;
}
