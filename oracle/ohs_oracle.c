/*
 * ohs_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See ohs_oracle.h for scope, citations and the pinning status
 * ("parity unpinned" for rustfft rounding and biquad/libm coefficient bits).
 *
 * Build with -ffp-contract=off: the reference is rustc output, which never
 * contracts a*b+c into an FMA, and the EQ recurrence is only reproducible
 * bit-for-bit with every product and sum rounded separately.
 */
#include "ohs_oracle.h"
#include "libm_musl.h"

#if defined(__x86_64__) || defined(__i386__)
#include <xmmintrin.h>
#endif

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BS OHS_OR_BLOCK_SIZE
#define NF OHS_OR_FFT_SIZE

/* ------------------------------------------------------------------------- */
/* complex helpers with num-complex 0.4.6 operation order                    */
/* ------------------------------------------------------------------------- */
typedef struct { float re, im; } cf32;

static inline cf32 c_mul(cf32 a, cf32 b)
{   /* (ar*br - ai*bi, ar*bi + ai*br) -- used at convolution.rs:271 */
    cf32 r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}
static inline cf32 c_add(cf32 a, cf32 b) { cf32 r = { a.re + b.re, a.im + b.im }; return r; }
static inline cf32 c_sub(cf32 a, cf32 b) { cf32 r = { a.re - b.re, a.im - b.im }; return r; }

/* ------------------------------------------------------------------------- */
/* FFT-1024: stand-in for rustfft 6.4.0 `Fft::process` (convolution.rs:56,   */
/* 88-90,117,129,255,276).  rustfft's contract: unnormalised, forward uses    */
/* e^{-2 pi i jk/N}; twiddles are computed in f64 and rounded to f32.  This   */
/* is a radix-4 decimation-in-time transform (rustfft's scalar planner also   */
/* picks a radix-4 family algorithm for 1024); its rounding is NOT claimed to */
/* equal rustfft's bit for bit (parity unpinned, see header).                 */
/* ------------------------------------------------------------------------- */
static cf32 g_tw[NF];          /* w^k = exp(-2 pi i k / 1024) */
static unsigned short g_rev4[NF];

__attribute__((constructor)) static void ohs_or_init_tables(void)
{
    for (int k = 0; k < NF; ++k) {
        double ang = -2.0 * 3.14159265358979323846264338327950288 * (double)k / (double)NF;
        g_tw[k].re = (float)cos(ang);
        g_tw[k].im = (float)sin(ang);
        /* base-4 digit reversal of a 5-digit index */
        int v = k, r = 0;
        for (int d = 0; d < 5; ++d) { r = (r << 2) | (v & 3); v >>= 2; }
        g_rev4[k] = (unsigned short)r;
    }
}

void ohs_or_fft1024(float *data, int inverse)
{
    cf32 *x = (cf32 *)data;
    for (int i = 0; i < NF; ++i) {
        int j = g_rev4[i];
        if (j > i) { cf32 t = x[i]; x[i] = x[j]; x[j] = t; }
    }
    for (int L = 4; L <= NF; L <<= 2) {
        const int q = L >> 2;
        const int step = NF / L;
        for (int j0 = 0; j0 < NF; j0 += L) {
            for (int k = 0; k < q; ++k) {
                cf32 w1 = g_tw[k * step], w2 = g_tw[2 * k * step], w3 = g_tw[3 * k * step];
                if (inverse) { w1.im = -w1.im; w2.im = -w2.im; w3.im = -w3.im; }
                cf32 a = x[j0 + k];
                cf32 b = c_mul(x[j0 + k + q], w1);
                cf32 c = c_mul(x[j0 + k + 2 * q], w2);
                cf32 d = c_mul(x[j0 + k + 3 * q], w3);
                cf32 apc = c_add(a, c), amc = c_sub(a, c);
                cf32 bpd = c_add(b, d), bmd = c_sub(b, d);
                /* forward: -j*(b-d) = (bmd.im, -bmd.re); inverse: +j*(b-d) */
                cf32 jb;
                if (!inverse) { jb.re = bmd.im; jb.im = -bmd.re; }
                else          { jb.re = -bmd.im; jb.im = bmd.re; }
                x[j0 + k]         = c_add(apc, bpd);
                x[j0 + k + q]     = c_add(amc, jb);
                x[j0 + k + 2 * q] = c_sub(apc, bpd);
                x[j0 + k + 3 * q] = c_sub(amc, jb);
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* small growable float FIFO = the reference's Vec<f32> + drain(..n)          */
/* ------------------------------------------------------------------------- */
typedef struct { float *p; size_t len, cap; } fvec;

static void fv_push(fvec *v, const float *src, size_t n)
{
    if (v->len + n > v->cap) {
        size_t nc = v->cap ? v->cap : 2 * BS;
        while (nc < v->len + n) nc *= 2;
        v->p = (float *)realloc(v->p, nc * sizeof(float));
        v->cap = nc;
    }
    if (n) memcpy(v->p + v->len, src, n * sizeof(float));
    v->len += n;
}
static void fv_pop_front(fvec *v, float *dst, size_t n)
{
    if (n) memcpy(dst, v->p, n * sizeof(float));
    memmove(v->p, v->p + n, (v->len - n) * sizeof(float));
    v->len -= n;
}
static void fv_copy(fvec *d, const fvec *s)
{
    d->p = NULL; d->len = d->cap = 0;
    fv_push(d, s->p, s->len);
}

/* ------------------------------------------------------------------------- */
/* ConvolutionPathData (convolution.rs:36-65)                                 */
/* ------------------------------------------------------------------------- */
typedef struct {
    size_t P;
    cf32 *ir_fft;       /* P x 1024 */
    cf32 *history;      /* P x 1024 ring */
    size_t history_index;
    float overlap[BS];
} path_t;

static void path_alloc(path_t *p, size_t P)
{
    p->P = P;
    p->ir_fft  = (cf32 *)calloc(P * NF, sizeof(cf32));
    p->history = (cf32 *)calloc(P * NF, sizeof(cf32));
    p->history_index = 0;
    memset(p->overlap, 0, sizeof(p->overlap));
}
static void path_release(path_t *p) { free(p->ir_fft); free(p->history); p->ir_fft = p->history = NULL; }

static void path_init_default(path_t *p)
{   /* convolution.rs:44-65: one partition, FFT of 512 zeros padded to 1024 */
    path_alloc(p, 1);
    ohs_or_fft1024((float *)p->ir_fft, 0);
}

struct ohs_or_engine {
    path_t paths[4];
    fvec in_l, in_r, out_l, out_r;           /* convolution.rs:76-79 */
    cf32 input_fft_buffer[NF];               /* :82 */
    cf32 conv_accumulator[NF];               /* :83 */
};

ohs_or_engine *ohs_or_engine_new(void)
{
    ohs_or_engine *e = (ohs_or_engine *)calloc(1, sizeof(*e));
    for (int i = 0; i < 4; ++i) path_init_default(&e->paths[i]);
    return e;
}

ohs_or_engine *ohs_or_engine_clone(const ohs_or_engine *s)
{
    ohs_or_engine *e = (ohs_or_engine *)calloc(1, sizeof(*e));
    for (int i = 0; i < 4; ++i) {
        const path_t *sp = &s->paths[i];
        path_alloc(&e->paths[i], sp->P);
        memcpy(e->paths[i].ir_fft, sp->ir_fft, sp->P * NF * sizeof(cf32));
        memcpy(e->paths[i].history, sp->history, sp->P * NF * sizeof(cf32));
        e->paths[i].history_index = sp->history_index;
        memcpy(e->paths[i].overlap, sp->overlap, sizeof(sp->overlap));
    }
    fv_copy(&e->in_l, &s->in_l);  fv_copy(&e->in_r, &s->in_r);
    fv_copy(&e->out_l, &s->out_l); fv_copy(&e->out_r, &s->out_r);
    memcpy(e->input_fft_buffer, s->input_fft_buffer, sizeof(e->input_fft_buffer));
    memcpy(e->conv_accumulator, s->conv_accumulator, sizeof(e->conv_accumulator));
    return e;
}

void ohs_or_engine_free(ohs_or_engine *e)
{
    if (!e) return;
    for (int i = 0; i < 4; ++i) path_release(&e->paths[i]);
    free(e->in_l.p); free(e->in_r.p); free(e->out_l.p); free(e->out_r.p);
    free(e);
}

size_t ohs_or_engine_num_partitions(const ohs_or_engine *e, int path) { return e->paths[path].P; }

void ohs_or_engine_set_ir(ohs_or_engine *e, int path, const float *ir, size_t len)
{   /* convolution.rs:111-139 */
    path_t *p = &e->paths[path];
    path_release(p);
    if (len == 0) {
        /* :114-118 one all-zero partition (FFT of zeros) */
        path_alloc(p, 1);
        ohs_or_fft1024((float *)p->ir_fft, 0);
    } else {
        size_t P = (len + BS - 1) / BS;             /* ir.chunks(512) */
        path_alloc(p, P);
        for (size_t i = 0; i < P; ++i) {
            size_t off = i * BS;
            size_t cnt = len - off < BS ? len - off : BS;
            cf32 *dst = p->ir_fft + i * NF;
            for (size_t k = 0; k < cnt; ++k) { dst[k].re = ir[off + k]; dst[k].im = 0.0f; }
            ohs_or_fft1024((float *)dst, 0);        /* :129 */
        }
    }
    /* :135-138 history = P zero spectra, index 0, overlap zeroed (path_alloc did all three);
       the engine-level FIFOs are untouched. */
}

/* convolution.rs:236-289 */
static void convolve_path_partitioned(ohs_or_engine *e, const float *input, path_t *p, float *output)
{
    cf32 *buf = e->input_fft_buffer;
    cf32 *acc = e->conv_accumulator;
    /* 1. real -> complex, zero-pad, forward FFT (:245-255) */
    for (int i = 0; i < BS; ++i) { buf[i].re = input[i]; buf[i].im = 0.0f; }
    for (int i = BS; i < NF; ++i) { buf[i].re = 0.0f; buf[i].im = 0.0f; }
    ohs_or_fft1024((float *)buf, 0);
    /* 2. store in history (:258) */
    memcpy(p->history + p->history_index * NF, buf, NF * sizeof(cf32));
    /* 3. accumulate over partitions, newest block with partition 0 (:261-273) */
    for (int j = 0; j < NF; ++j) { acc[j].re = 0.0f; acc[j].im = 0.0f; }
    const size_t P = p->P;
    for (size_t i = 0; i < P; ++i) {
        size_t hidx = (p->history_index + P - i) % P;
        const cf32 *x = p->history + hidx * NF;
        const cf32 *h = p->ir_fft + i * NF;
        for (int j = 0; j < NF; ++j) acc[j] = c_add(acc[j], c_mul(x[j], h[j]));
    }
    /* 4. inverse FFT, unnormalised (:276) */
    ohs_or_fft1024((float *)acc, 1);
    /* 5. overlap-add (:279-284) */
    const float scale = 1.0f / (float)NF;
    for (int i = 0; i < BS; ++i) {
        output[i] = acc[i].re * scale + p->overlap[i];
        p->overlap[i] = acc[i + BS].re * scale;
    }
    /* 6. (:286) */
    p->history_index = (p->history_index + 1) % P;
}

/* convolution.rs:184-233 */
static void process_internal_block(ohs_or_engine *e, const float *in_l, const float *in_r,
                                   float *out_l, float *out_r)
{
    float lsl[BS], lsr[BS], rsl[BS], rsr[BS];
    convolve_path_partitioned(e, in_l, &e->paths[OHS_OR_LSL], lsl);
    convolve_path_partitioned(e, in_l, &e->paths[OHS_OR_LSR], lsr);
    convolve_path_partitioned(e, in_r, &e->paths[OHS_OR_RSL], rsl);
    convolve_path_partitioned(e, in_r, &e->paths[OHS_OR_RSR], rsr);
    for (int i = 0; i < BS; ++i) {
        out_l[i] = lsl[i] + rsl[i];
        out_r[i] = lsr[i] + rsr[i];
    }
}

/* convolution.rs:141-182 */
void ohs_or_engine_process_block(ohs_or_engine *e, const float *in_l, const float *in_r,
                                 float *out_l, float *out_r, size_t n)
{
    fv_push(&e->in_l, in_l, n);
    fv_push(&e->in_r, in_r, n);
    while (e->in_l.len >= BS) {
        float cl[BS], cr[BS], pl[BS], pr[BS];
        fv_pop_front(&e->in_l, cl, BS);
        fv_pop_front(&e->in_r, cr, BS);
        process_internal_block(e, cl, cr, pl, pr);
        fv_push(&e->out_l, pl, BS);
        fv_push(&e->out_r, pr, BS);
    }
    if (e->out_l.len >= n) {
        fv_pop_front(&e->out_l, out_l, n);
        fv_pop_front(&e->out_r, out_r, n);
    } else {
        /* :176-181 silence fallback; FIFOs keep their content */
        for (size_t i = 0; i < n; ++i) { out_l[i] = 0.0f; out_r[i] = 0.0f; }
    }
}

/* ------------------------------------------------------------------------- */
/* biquad 0.4.2 Coefficients::<f32>::from_params (call sites                  */
/* parametric_eq.rs:64-70,105-111), restated from the published crate.        */
/* All arithmetic is f32, evaluated left to right as the crate writes it.     */
/* libm 0.1.4's sinf/cosf/powf (a port of musl's) are restated in             */
/* oracle/libm_musl.h, not taken from this platform's libm (parity unpinned   */
/* at the last-ulp level all the same: the crate is not on this machine).     */
/* ------------------------------------------------------------------------- */
int ohs_or_biquad_coeffs(int filter_type, float fs, float fc, float q, float gain_db, float out[5])
{
    if (2.0f * fc > fs) return 1;      /* Errors::OutsideNyquist */
    if (q < 0.0f) return 2;            /* Errors::NegativeQ */

    const float pi = 3.14159265358979323846264338327950288f;
    const float omega = 2.0f * pi * fc / fs;
    const float omega_s = ohs_musl_sinf(omega);
    const float omega_c = ohs_musl_cosf(omega);
    const float alpha = omega_s / (2.0f * q);
    float b0, b1, b2, a0, a1, a2;

    switch (filter_type) {
    case OHS_OR_LOWPASS:
        b0 = (1.0f - omega_c) * 0.5f; b1 = 1.0f - omega_c; b2 = (1.0f - omega_c) * 0.5f;
        a0 = 1.0f + alpha; a1 = -2.0f * omega_c; a2 = 1.0f - alpha;
        break;
    case OHS_OR_HIGHPASS:
        b0 = (1.0f + omega_c) * 0.5f; b1 = -(1.0f + omega_c); b2 = (1.0f + omega_c) * 0.5f;
        a0 = 1.0f + alpha; a1 = -2.0f * omega_c; a2 = 1.0f - alpha;
        break;
    case OHS_OR_BANDPASS:
        b0 = omega_s / 2.0f; b1 = 0.0f; b2 = -(omega_s / 2.0f);
        a0 = 1.0f + alpha; a1 = -2.0f * omega_c; a2 = 1.0f - alpha;
        break;
    case OHS_OR_NOTCH:
        b0 = 1.0f; b1 = -2.0f * omega_c; b2 = 1.0f;
        a0 = 1.0f + alpha; a1 = -2.0f * omega_c; a2 = 1.0f - alpha;
        break;
    case OHS_OR_ALLPASS:
        b0 = 1.0f - alpha; b1 = -2.0f * omega_c; b2 = 1.0f + alpha;
        a0 = 1.0f + alpha; a1 = -2.0f * omega_c; a2 = 1.0f - alpha;
        break;
    case OHS_OR_LOWSHELF: {
        const float a = ohs_musl_powf(10.0f, gain_db / 40.0f);
        const float sa = sqrtf(a);
        b0 = a * ((a + 1.0f) - (a - 1.0f) * omega_c + 2.0f * alpha * sa);
        b1 = 2.0f * a * ((a - 1.0f) - (a + 1.0f) * omega_c);
        b2 = a * ((a + 1.0f) - (a - 1.0f) * omega_c - 2.0f * alpha * sa);
        a0 = (a + 1.0f) + (a - 1.0f) * omega_c + 2.0f * alpha * sa;
        a1 = -2.0f * ((a - 1.0f) + (a + 1.0f) * omega_c);
        a2 = (a + 1.0f) + (a - 1.0f) * omega_c - 2.0f * alpha * sa;
        break; }
    case OHS_OR_HIGHSHELF: {
        const float a = ohs_musl_powf(10.0f, gain_db / 40.0f);
        const float sa = sqrtf(a);
        b0 = a * ((a + 1.0f) + (a - 1.0f) * omega_c + 2.0f * alpha * sa);
        b1 = -2.0f * a * ((a - 1.0f) + (a + 1.0f) * omega_c);
        b2 = a * ((a + 1.0f) + (a - 1.0f) * omega_c - 2.0f * alpha * sa);
        a0 = (a + 1.0f) - (a - 1.0f) * omega_c + 2.0f * alpha * sa;
        a1 = 2.0f * ((a - 1.0f) - (a + 1.0f) * omega_c);
        a2 = (a + 1.0f) - (a - 1.0f) * omega_c - 2.0f * alpha * sa;
        break; }
    case OHS_OR_PEAK:
    default: {
        const float a = ohs_musl_powf(10.0f, gain_db / 40.0f);
        b0 = 1.0f + alpha * a; b1 = -2.0f * omega_c; b2 = 1.0f - alpha * a;
        a0 = 1.0f + alpha / a; a1 = -2.0f * omega_c; a2 = 1.0f - alpha / a;
        break; }
    }
    out[0] = b0 / a0; out[1] = b1 / a0; out[2] = b2 / a0; out[3] = a1 / a0; out[4] = a2 / a0;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* BiquadFilter / StereoParametricEQ (parametric_eq.rs:46-210)                */
/* ------------------------------------------------------------------------- */
typedef struct {
    float s1, s2;               /* DirectForm2Transposed state */
    float b0, b1, b2, a1, a2;   /* Coefficients */
    int enabled;
} biquad_t;

struct ohs_or_eq {
    size_t num_bands;
    biquad_t *left, *right;
};

static void biquad_init(biquad_t *b, float fs)
{   /* parametric_eq.rs:63-76: PeakingEQ(0 dB) @ 20 Hz, Q 0.707, disabled */
    float c[5];
    ohs_or_biquad_coeffs(OHS_OR_PEAK, fs, 20.0f, 0.707f, 0.0f, c);
    b->s1 = b->s2 = 0.0f;
    b->b0 = c[0]; b->b1 = c[1]; b->b2 = c[2]; b->a1 = c[3]; b->a2 = c[4];
    b->enabled = 0;
}

static inline float biquad_process_sample(biquad_t *b, float x)
{   /* parametric_eq.rs:116-122 -> biquad 0.4 DirectForm2Transposed::run */
    if (!b->enabled) return x;
    float out = b->s1 + b->b0 * x;
    b->s1 = b->s2 + b->b1 * x - b->a1 * out;
    b->s2 = b->b2 * x - b->a2 * out;
    return out;
}

ohs_or_eq *ohs_or_eq_new(size_t num_bands, float fs)
{
    ohs_or_eq *eq = (ohs_or_eq *)calloc(1, sizeof(*eq));
    eq->num_bands = num_bands;
    eq->left  = (biquad_t *)calloc(num_bands ? num_bands : 1, sizeof(biquad_t));
    eq->right = (biquad_t *)calloc(num_bands ? num_bands : 1, sizeof(biquad_t));
    for (size_t i = 0; i < num_bands; ++i) { biquad_init(&eq->left[i], fs); biquad_init(&eq->right[i], fs); }
    return eq;
}
void ohs_or_eq_free(ohs_or_eq *eq) { if (eq) { free(eq->left); free(eq->right); free(eq); } }

int ohs_or_eq_update_band(ohs_or_eq *eq, size_t band, float fs, int filter_type,
                          float fc, float q, float gain_db, int enabled)
{   /* parametric_eq.rs:144-164; idx >= num_bands is silently ignored */
    if (band >= eq->num_bands) return 0;
    float c[5];
    int rc = ohs_or_biquad_coeffs(filter_type, fs, fc, q, gain_db, c);
    if (rc) return rc;          /* the reference panics here (unwrap) */
    biquad_t *bl = &eq->left[band], *br = &eq->right[band];
    bl->b0 = c[0]; bl->b1 = c[1]; bl->b2 = c[2]; bl->a1 = c[3]; bl->a2 = c[4]; bl->enabled = enabled != 0;
    br->b0 = c[0]; br->b1 = c[1]; br->b2 = c[2]; br->a1 = c[3]; br->a2 = c[4]; br->enabled = enabled != 0;
    return 0;
}

void ohs_or_eq_set_band_coeffs(ohs_or_eq *eq, size_t band, const float c[5], int enabled)
{
    if (band >= eq->num_bands) return;
    biquad_t *bl = &eq->left[band], *br = &eq->right[band];
    bl->b0 = c[0]; bl->b1 = c[1]; bl->b2 = c[2]; bl->a1 = c[3]; bl->a2 = c[4]; bl->enabled = enabled != 0;
    br->b0 = c[0]; br->b1 = c[1]; br->b2 = c[2]; br->a1 = c[3]; br->a2 = c[4]; br->enabled = enabled != 0;
}

void ohs_or_eq_get_band_coeffs(const ohs_or_eq *eq, size_t band, float c[5], int *enabled)
{
    const biquad_t *b = &eq->left[band];
    c[0] = b->b0; c[1] = b->b1; c[2] = b->b2; c[3] = b->a1; c[4] = b->a2;
    if (enabled) *enabled = b->enabled;
}

void ohs_or_eq_process_block(ohs_or_eq *eq, float *l, float *r, size_t n)
{   /* parametric_eq.rs:166-179: sample-major, band-minor, L then R per band */
    for (size_t i = 0; i < n; ++i) {
        float sl = l[i], sr = r[i];
        for (size_t j = 0; j < eq->num_bands; ++j) {
            sl = biquad_process_sample(&eq->left[j], sl);
            sr = biquad_process_sample(&eq->right[j], sr);
        }
        l[i] = sl; r[i] = sr;
    }
}

void ohs_or_eq_reset(ohs_or_eq *eq)
{   /* parametric_eq.rs:181-188 */
    for (size_t j = 0; j < eq->num_bands; ++j) {
        eq->left[j].s1 = eq->left[j].s2 = 0.0f;
        eq->right[j].s1 = eq->right[j].s2 = 0.0f;
    }
}

void ohs_or_eq_frequency_response(const ohs_or_eq *eq, float fs, const float *freqs, size_t n, float *out)
{   /* parametric_eq.rs:190-209 with num-complex 0.4.6 operation order */
    const float pi = 3.14159265358979323846264338327950288f;
    for (size_t i = 0; i < n; ++i) {
        cf32 resp = { 1.0f, 0.0f };
        for (size_t j = 0; j < eq->num_bands; ++j) {
            const biquad_t *b = &eq->left[j];
            if (!b->enabled) continue;
            float omega = 2.0f * pi * freqs[i] / fs;
            cf32 z = { 1.0f * cosf(-omega), 1.0f * sinf(-omega) };      /* from_polar */
            float ns = z.re * z.re + z.im * z.im;                       /* inv() */
            cf32 zi = { z.re / ns, -z.im / ns };                        /* z.powi(-1) */
            cf32 zi2 = c_mul(zi, zi);                                   /* z.powi(-2) */
            cf32 t1 = { b->b1 * zi.re, b->b1 * zi.im };
            cf32 t2 = { b->b2 * zi2.re, b->b2 * zi2.im };
            cf32 num = { b->b0 + t1.re, t1.im };
            num = c_add(num, t2);
            cf32 u1 = { b->a1 * zi.re, b->a1 * zi.im };
            cf32 u2 = { b->a2 * zi2.re, b->a2 * zi2.im };
            cf32 den = { 1.0f + u1.re, u1.im };
            den = c_add(den, u2);
            float dn = den.re * den.re + den.im * den.im;               /* Div */
            cf32 r;
            r.re = (num.re * den.re + num.im * den.im) / dn;
            r.im = (num.im * den.re - num.re * den.im) / dn;
            resp = c_mul(resp, r);
        }
        out[i] = hypotf(resp.re, resp.im);                              /* norm() */
    }
}

/* ------------------------------------------------------------------------- */
/* Plugin::process (lib.rs:1169-1207)                                         */
/* ------------------------------------------------------------------------- */
void ohs_or_chain_process(ohs_or_engine *e, ohs_or_eq *eq, int bypass, int eq_enable,
                          float gain, float *l, float *r, size_t n)
{
    if (bypass) return;                                      /* :1169 */
    if (eq_enable) ohs_or_eq_process_block(eq, l, r, n);     /* :1179-1194 (coeff refresh is the caller's) */
    float *il = (float *)malloc((n ? n : 1) * sizeof(float));
    float *ir = (float *)malloc((n ? n : 1) * sizeof(float));
    memcpy(il, l, n * sizeof(float));                        /* :1197-1198 */
    memcpy(ir, r, n * sizeof(float));
    ohs_or_engine_process_block(e, il, ir, l, r, n);         /* :1199 */
    for (size_t i = 0; i < n; ++i) { l[i] *= gain; r[i] *= gain; }   /* :1202-1207 */
    free(il); free(ir);
}

/* ------------------------------------------------------------------------- */
/* f64 ground truth                                                           */
/* ------------------------------------------------------------------------- */
void ohs_or_direct_conv_f64(const float *x, size_t n, const float *h, size_t taps, double *y)
{
    for (size_t i = 0; i < n; ++i) {
        double acc = 0.0;
        size_t kmax = i + 1 < taps ? i + 1 : taps;
        for (size_t k = 0; k < kmax; ++k) acc += (double)h[k] * (double)x[i - k];
        y[i] = acc;
    }
}

static void conv_acc_f64(const double *x, size_t n, const float *h, size_t taps, double *y)
{
    for (size_t i = 0; i < n; ++i) {
        double acc = 0.0;
        size_t kmax = i + 1 < taps ? i + 1 : taps;
        for (size_t k = 0; k < kmax; ++k) acc += (double)h[k] * x[i - k];
        y[i] += acc;
    }
}

void ohs_or_binaural_f64(const double *x_l, const double *x_r, size_t n,
                         const float *h0, size_t t0, const float *h1, size_t t1,
                         const float *h2, size_t t2, const float *h3, size_t t3,
                         double *out_l, double *out_r)
{
    for (size_t i = 0; i < n; ++i) { out_l[i] = 0.0; out_r[i] = 0.0; }
    conv_acc_f64(x_l, n, h0, t0, out_l);   /* Lsl */
    conv_acc_f64(x_r, n, h2, t2, out_l);   /* Rsl */
    conv_acc_f64(x_l, n, h1, t1, out_r);   /* Lsr */
    conv_acc_f64(x_r, n, h3, t3, out_r);   /* Rsr */
}

void ohs_or_eq_f64(const float *coeffs, const int *enabled, size_t nb,
                   const float *x, size_t n, double *state, double *y)
{
    for (size_t i = 0; i < n; ++i) {
        double s = (double)x[i];
        for (size_t j = 0; j < nb; ++j) {
            if (!enabled[j]) continue;
            const float *c = coeffs + 5 * j;
            double out = state[2 * j] + (double)c[0] * s;
            state[2 * j]     = state[2 * j + 1] + (double)c[1] * s - (double)c[3] * out;
            state[2 * j + 1] = (double)c[2] * s - (double)c[4] * out;
            s = out;
        }
        y[i] = s;
    }
}

/* ------------------------------------------------------------------------- */
/* Denormal handling of the CALLING THREAD, as the shipped plugin runs:       */
/* nih-plug wraps Plugin::process (src/lib.rs:1156) in a flush-to-zero scope  */
/* (MXCSR.FTZ: denormal RESULTS become signed zeros; denormal inputs are      */
/* still read).  mode 0 = IEEE (what the reference's unit tests run with),    */
/* 1 = FTZ, 2 = FTZ | DAZ.  Returns the previous mode.  The arithmetic of     */
/* this file is scalar / vector SSE, which MXCSR governs.                     */
/* ------------------------------------------------------------------------- */
int ohs_or_set_flush_denormals(int mode)
{
#if defined(__x86_64__) || defined(__i386__)
    const unsigned csr = _mm_getcsr();
    const int prev = (csr & 0x8000u) ? ((csr & 0x0040u) ? 2 : 1) : 0;
    unsigned next = csr & ~(0x8000u | 0x0040u);
    if (mode >= 1) next |= 0x8000u;     /* FTZ */
    if (mode >= 2) next |= 0x0040u;     /* DAZ */
    _mm_setcsr(next);
    return prev;
#else
    (void)mode;
    return -1;
#endif
}
