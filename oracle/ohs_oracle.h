/*
 * ohs_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the Open-Headstage `src/dsp` hot path, used only as
 * the parity checker by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Nothing under open_headstage_amd/ may include, link or
 * call it.
 *
 * What it follows (paths relative to /root/reference):
 *   src/dsp/convolution.rs:22-23,28-33,44-65,87-108,111-139,141-182,184-233,236-289
 *   src/dsp/parametric_eq.rs:23-35,37-44,62-123,131-188,190-209
 *   src/lib.rs:1169-1207                        (EQ -> convolution -> gain order)
 *
 * Third-party arithmetic that is NOT vendored under /root/reference and is
 * restated here from the published algorithms of the pinned crates:
 *   rustfft 6.4.0      (Cargo.lock:2579-2580)  unnormalised forward / inverse DFT-1024
 *   num-complex 0.4.6  (Cargo.lock:1812-1813)  complex mul / add / div / powi / norm
 *   biquad 0.4.2       (Cargo.lock:314-315)    RBJ-cookbook Coefficients::from_params,
 *                                              DirectForm2Transposed::run
 *   libm 0.1.4         (Cargo.lock:1521-1522)  sinf / cosf / powf / sqrtf used by biquad
 *
 * PINNING STATUS
 *   pinned   : the reference's own five unit tests for this path (three
 *              convolution known-answer tests convolution.rs:317-421, two EQ
 *              tests parametric_eq.rs:218-238) are re-created verbatim in
 *              tests/test_oracle_reference_kats.py and pass against this file.
 *   UNPINNED : (parity unpinned) the f32 rounding of rustfft's FFT-1024 and
 *              the numeric value of biquad/libm coefficients.  The reference
 *              holds no golden vectors for them and no Rust toolchain exists
 *              in the build container, so they cannot be generated.  The
 *              <=1e-6 RMS criterion is therefore evaluated GPU-vs-this-file
 *              and both-vs-the-f64 ground truth below.
 */
#ifndef OHS_ORACLE_H
#define OHS_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OHS_OR_BLOCK_SIZE 512   /* convolution.rs:22 */
#define OHS_OR_FFT_SIZE   1024  /* convolution.rs:23 */

/* ConvolutionPath discriminants, convolution.rs:28-33 */
enum { OHS_OR_LSL = 0, OHS_OR_LSR = 1, OHS_OR_RSL = 2, OHS_OR_RSR = 3 };

/* FilterType discriminants, parametric_eq.rs:23-35 */
enum {
    OHS_OR_PEAK = 0, OHS_OR_LOWSHELF = 1, OHS_OR_HIGHSHELF = 2, OHS_OR_LOWPASS = 3,
    OHS_OR_HIGHPASS = 4, OHS_OR_BANDPASS = 5, OHS_OR_NOTCH = 6, OHS_OR_ALLPASS = 7
};

/* ---- f32 restatement of rustfft's contract: in-place complex DFT-1024 ------- */
/* data = 1024 interleaved (re,im) pairs; inverse!=0 -> e^{+...}; unnormalised.  */
void ohs_or_fft1024(float *data, int inverse);

/* ---- ConvolutionEngine (convolution.rs:68-295) -------------------------------*/
typedef struct ohs_or_engine ohs_or_engine;
ohs_or_engine *ohs_or_engine_new(void);                         /* :87-108  */
ohs_or_engine *ohs_or_engine_clone(const ohs_or_engine *e);     /* :68 derive(Clone) */
void ohs_or_engine_free(ohs_or_engine *e);
void ohs_or_engine_set_ir(ohs_or_engine *e, int path, const float *ir, size_t len); /* :111-139 */
void ohs_or_engine_process_block(ohs_or_engine *e, const float *in_l, const float *in_r,
                                 float *out_l, float *out_r, size_t n);             /* :141-182 */
size_t ohs_or_engine_num_partitions(const ohs_or_engine *e, int path); /* test :395-399 */

/* ---- StereoParametricEQ (parametric_eq.rs:125-210) ---------------------------*/
typedef struct ohs_or_eq ohs_or_eq;
/* biquad::Coefficients::from_params restated; out = {b0,b1,b2,a1,a2} normalised
 * by a0.  Returns 0, or 1 = OutsideNyquist, 2 = NegativeQ (the reference
 * unwrap()s, i.e. panics, on those: parametric_eq.rs:105-111).                   */
int ohs_or_biquad_coeffs(int filter_type, float fs, float fc, float q, float gain_db,
                         float out[5]);
ohs_or_eq *ohs_or_eq_new(size_t num_bands, float fs);           /* :132-142 */
void ohs_or_eq_free(ohs_or_eq *eq);
int  ohs_or_eq_update_band(ohs_or_eq *eq, size_t band, float fs, int filter_type,
                           float fc, float q, float gain_db, int enabled); /* :144-164 */
void ohs_or_eq_set_band_coeffs(ohs_or_eq *eq, size_t band, const float c[5], int enabled);
void ohs_or_eq_get_band_coeffs(const ohs_or_eq *eq, size_t band, float c[5], int *enabled);
void ohs_or_eq_process_block(ohs_or_eq *eq, float *l, float *r, size_t n); /* :166-179 */
void ohs_or_eq_reset(ohs_or_eq *eq);                            /* :181-188 */
void ohs_or_eq_frequency_response(const ohs_or_eq *eq, float fs, const float *freqs,
                                  size_t n, float *out);        /* :190-209 */

/* ---- Plugin::process chain (lib.rs:1169-1207) --------------------------------*/
/* in place on l,r: [eq if eq_enable] -> conv -> *= gain; bypass -> untouched.    */
void ohs_or_chain_process(ohs_or_engine *e, ohs_or_eq *eq, int bypass, int eq_enable,
                          float gain, float *l, float *r, size_t n);

/* ---- f64 ground truth (the build's own, NOT from the reference) --------------*/
/* y[i] = sum_k h[k] x[i-k], i<n, double accumulation                            */
void ohs_or_direct_conv_f64(const float *x, size_t n, const float *h, size_t taps, double *y);
/* out_l = x_l*h[0] + x_r*h[2];  out_r = x_l*h[1] + x_r*h[3]   (convolution.rs:228-231) */
void ohs_or_binaural_f64(const double *x_l, const double *x_r, size_t n,
                         const float *h0, size_t t0, const float *h1, size_t t1,
                         const float *h2, size_t t2, const float *h3, size_t t3,
                         double *out_l, double *out_r);
/* DF2T cascade in double with the given (f32-rounded) coefficients; state[2*nb]  */
void ohs_or_eq_f64(const float *coeffs /* nb*5 */, const int *enabled, size_t nb,
                   const float *x, size_t n, double *state, double *y);

/* flush-to-zero mode of the calling thread (0 = IEEE, 1 = FTZ, 2 = FTZ | DAZ); returns the previous mode, -1 where unsupported */
int ohs_or_set_flush_denormals(int mode);

#ifdef __cplusplus
}
#endif
#endif
