/*
 * ohs_hip.h -- C ABI of the MI355X-native binaural convolution core.
 *
 * Drop-in boundary for Open-Headstage's `src/dsp` hot path.  Every entry point
 * names the reference interface it replaces (paths relative to the reference
 * repository).  The reference host is Rust; the binding a maintainer adds is in
 * INTEGRATION.md (extern "C" block + a shim keeping `ConvolutionEngine` /
 * `StereoParametricEQ` source-compatible with src/lib.rs).
 *
 * Conventions
 *   - plain pointers and sizes only; `int` status, 0 == OHS_OK; nothing throws
 *     or unwinds across this boundary.
 *   - one handle == one thread at a time (the reference takes `&mut self`,
 *     convolution.rs:111,141; parametric_eq.rs:144,166).
 *   - the library owns device memory; the caller owns every host buffer, which
 *     need only stay valid for the duration of the call.
 *   - engine / eq / chain calls are blocking: outputs are complete on return.
 *   - there is NO CPU fallback: without a usable HIP device every create call
 *     fails with OHS_ERR_NO_DEVICE.
 *   - audio is planar f32, as nih-plug hands it to Plugin::process
 *     (src/lib.rs:1175-1177).
 */
#ifndef OHS_HIP_H
#define OHS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OHS_BLOCK_SIZE 512   /* src/dsp/convolution.rs:22 BLOCK_SIZE */
#define OHS_FFT_SIZE   1024  /* src/dsp/convolution.rs:23 FFT_SIZE   */
#define OHS_MAX_EQ_BANDS 64  /* num_bands is the caller's choice (parametric_eq.rs:132); the plugin uses 10 (lib.rs:40) */

/* status codes */
enum {
    OHS_OK = 0,
    OHS_ERR_INVALID_ARG = 1,
    OHS_ERR_NO_DEVICE = 2,        /* no HIP device / runtime unusable: no CPU fallback exists */
    OHS_ERR_HIP = 3,              /* a HIP runtime call failed; see ohs_last_error() */
    OHS_ERR_OUTSIDE_NYQUIST = 4,  /* biquad::Errors::OutsideNyquist (the reference unwrap()s: parametric_eq.rs:111) */
    OHS_ERR_NEGATIVE_Q = 5,       /* biquad::Errors::NegativeQ */
    OHS_ERR_ALLOC = 6,
    OHS_ERR_IO = 7,               /* a file could not be opened / read */
    OHS_ERR_PARSE = 8,            /* malformed CSV / unsupported filter type; see ohs_last_error() */
    OHS_ERR_RCCL = 9              /* librccl.so could not be loaded, or an RCCL call failed; see ohs_last_error() */
};

/* ConvolutionPath discriminants -- src/dsp/convolution.rs:28-33.
 * L/R speaker -> l/r ear; out_l = Lsl + Rsl, out_r = Lsr + Rsr (:228-231). */
enum { OHS_PATH_LSL = 0, OHS_PATH_LSR = 1, OHS_PATH_RSL = 2, OHS_PATH_RSR = 3 };

/* FilterType discriminants -- src/dsp/parametric_eq.rs:23-35 (also the CLAP enum values). */
enum {
    OHS_FILTER_PEAK = 0, OHS_FILTER_LOWSHELF = 1, OHS_FILTER_HIGHSHELF = 2, OHS_FILTER_LOWPASS = 3,
    OHS_FILTER_HIGHPASS = 4, OHS_FILTER_BANDPASS = 5, OHS_FILTER_NOTCH = 6, OHS_FILTER_ALLPASS = 7
};

/* ---- library ---------------------------------------------------------------- */
int         ohs_version(void);               /* (major<<16)|(minor<<8)|patch */
const char *ohs_status_string(int status);
const char *ohs_last_error(void);            /* thread-local detail of the last failure */
int         ohs_device_count(int *count);    /* OHS_ERR_NO_DEVICE if the runtime is unusable */
/* "domain:bus:device.function" of HIP device `device` (hipDeviceGetPCIBusId): which physical GPU a shard ran on */
int         ohs_device_pci_bus_id(int device, char *out, size_t capacity);

/* Diagnostic: `batch` natural-order complex DFT-1024s (interleaved re,im; host pointers) through the
 * wavefront FFT primitive -- the device counterpart of rustfft's `Fft::process`
 * (convolution.rs:129,255,276): unnormalised, forward = e^{-2 pi i jk/N}. */
int         ohs_debug_fft1024(int device, const float *in, float *out, int inverse, size_t batch);

/* ---- ConvolutionEngine  (src/dsp/convolution.rs:68-295) ---------------------- */
typedef struct ohs_engine ohs_engine;

/* ConvolutionEngine::new()  convolution.rs:87-108.  Four default paths, each one
 * all-zero partition => silence (:46-63). */
int  ohs_engine_create(int device, ohs_engine **out);
/* #[derive(Clone)]  convolution.rs:68 -- deep copy of IR spectra, history, overlap, FIFOs. */
int  ohs_engine_clone(const ohs_engine *src, ohs_engine **out);
/* Drop */
void ohs_engine_destroy(ohs_engine *e);
/* ConvolutionEngine::set_ir(&mut self, path, &[f32])  convolution.rs:111-139.
 * len == 0 => one zero partition (mute, :114-118); else ceil(len/512) partitions,
 * each chunk zero-padded to 1024 and forward-transformed (:120-132); resets THAT
 * path's history, history index and overlap (:135-138); FIFOs untouched. */
int  ohs_engine_set_ir(ohs_engine *e, int path, const float *ir, size_t len);
/* ConvolutionEngine::process_block(&mut self, in_l, in_r, out_l, out_r)
 * convolution.rs:141-182.  All four slices have n elements.  Appends to the input
 * FIFOs, runs every complete 512-block, pops n outputs, or fills zeros when fewer
 * than n are available (:176-181).
 * Responses of more than one partition (convolution.rs:261-273): the part of a block's sum over the partitions that
 * pairs with blocks processed EARLIER does not wait for the call -- it is computed behind the previous call (a kernel
 * queued behind the call's own; in real-time mode the resident kernel, between the calls), so a 1 024-frame call on
 * 4 x 16 384 taps costs what it costs on 4 x 512 (+ 2 us).  Nothing to do for the host; set_ir and calls of more than
 * 4 096 frames simply start over.  A host that calls back to back (an offline render) is served fastest with real-time
 * mode OFF: the resident kernel's one workgroup does that work between the calls. */
int  ohs_engine_process_block(ohs_engine *e, const float *in_l, const float *in_r,
                              float *out_l, float *out_r, size_t n);
/* engine.paths[p].ir_fft_partitions.len()  (asserted by the reference test, convolution.rs:395-399) */
int  ohs_engine_num_partitions(const ohs_engine *e, int path, size_t *out);
/* Real-time mode (no reference counterpart -- the reference runs on the audio
 * thread's CPU).  enable != 0 starts a RESIDENT kernel (one workgroup) that serves ohs_engine_process_block --
 * and ohs_chain_process with up to 12 enabled EQ bands -- through a mailbox in pinned host memory: a call that
 * completes at most four 512-blocks (host blocks <= 2048 frames) launches nothing, allocates nothing and costs
 * one doorbell round trip + one block's computation.  Results are bit-identical to the launch-per-call path,
 * which serves every other call.  The kernel leaves the GPU when real-time mode is switched off, on set_ir /
 * clone / destroy / any call it cannot serve (it is restarted by the next call it can), and by itself after
 * ~2 s without a call.  It holds one CU while it lives: a device runs at most num_CUs / 4 of them
 * (ohs_rt_configure), further engines in real-time mode are served launch per call until a slot frees up.
 * Whenever ANY handle of this library on the same device is about to free device memory or wait for the whole
 * device (set_ir, destroy, reset, buffer growth), it asks the device's resident kernels to leave first and holds new
 * ones off until it is done -- a resident kernel would otherwise stall such a wait for as long as it lives; a call
 * arriving meanwhile is served launch per call, the next one restarts the kernel.
 * Same threading rule as every handle: one thread at a time. */
int  ohs_engine_set_realtime(ohs_engine *e, int enable);
/* The two deployment options of the resident kernels, process-wide, for kernels started afterwards:
 * max_resident_per_device (< 0: the default, a quarter of the device's CUs; 0: none -- every engine in real-time mode
 * is served launch per call) and idle_polls, the doorbell polls (~0.6 us each) after which an idle kernel leaves
 * (0: keep the current value; default 3 000 000 ~ 2 s).  The library reads NO environment variable: these are the only
 * run-time settings that are not per handle. */
int  ohs_rt_configure(int max_resident_per_device, uint64_t idle_polls);
/* Delay the FIFO adapter of process_block adds for a constant host block size n (convolution.rs:163-181):
 * 0 when n is a multiple of 512; otherwise the frames answered with silence before the output FIFO can
 * serve every call (e.g. 256 for n = 256).  What a plugin host reports as its latency for compensation. */
int  ohs_engine_latency_samples(const ohs_engine *e, size_t n, size_t *latency);
/* Diagnostics of the resident kernel: 100 MHz device timestamps of the last call it served -- doorbell seen,
 * inputs visible (after the acquire), body done, every wave's stores drained, released; out[5] = shader-clock
 * cycles between the first and the last (their ratio is the clock the wave really ran at); out[8..14] = the
 * same clock inside the call: entry, FIFO pop done, inputs transformed, barrier, MAC done, inverse FFTs done,
 * outputs stored (wave 0 only). */
int  ohs_engine_debug_rt_stamps(const ohs_engine *e, uint64_t out[16]);
/* Frames this engine has answered with that silence so far = its current delay in frames. */
int  ohs_engine_silent_frames(const ohs_engine *e, uint64_t *frames);
/* Denormal handling of every kernel this handle launches: the reference's unit tests run
 * with IEEE denormals, but the shipped plugin's Plugin::process (src/lib.rs:1156) runs inside nih-plug's
 * flush-to-zero scope.  mode 0 (default): IEEE, what the parity tests compare.  mode 1: results that would be
 * denormal are flushed to signed zero, denormal inputs are still read (x86 MXCSR.FTZ, what nih-plug sets).  mode 2:
 * denormal inputs are read as zero as well (FTZ | DAZ, what many hosts set on their audio threads).  Set through the
 * MODE register of each kernel.  In modes 1 and 2 the EQ stays bit-exact against the reference arithmetic run in the
 * same mode (tests/test_gpu_ftz.py) up to the default (ring) kernel's documented corners, which flushing makes
 * reachable: a result flushed to -0.0 comes out as +0.0, and a denormal INPUT sample is read as zero also in mode 1
 * (ohs_eq_set_exact_specials selects the kernel without either).
 * Also available as ohs_eq_set_flush_denormals and ohs_batch_set_flush_denormals; clones inherit the mode. */
int  ohs_engine_set_flush_denormals(ohs_engine *e, int mode);

/* ---- StereoParametricEQ  (src/dsp/parametric_eq.rs:125-210) ------------------ */
typedef struct ohs_eq ohs_eq;

/* biquad::Coefficients::<f32>::from_params as called at parametric_eq.rs:64-70,105-111;
 * out = {b0,b1,b2,a1,a2} normalised by a0.  gain_db is used by Peak/LowShelf/HighShelf only.
 * A CONVENIENCE, NOT A PARITY PATH: the formulas are the crate's, and sinf / cosf / powf are a restatement of the
 * `libm` crate 0.1.4 the reference links (a port of musl's routines; csrc/libm_musl.h) rather than the host's libm,
 * so the result is the same on every host -- but whether it equals the crate's last bit cannot be checked here
 * (parity unpinned), and ONE ulp in ONE coefficient moves the EQ's output by up to 2e-4 relative RMS.  A host that
 * must match the reference to 1e-6 computes the coefficients with the `biquad` crate and passes them to
 * ohs_eq_set_band_coeffs / ohs_batch_set_eq_band_coeffs. */
int  ohs_biquad_coeffs(int filter_type, float fs, float fc, float q, float gain_db, float out[5]);
/* StereoParametricEQ::new(num_bands, fs)  parametric_eq.rs:132-142: every band
 * PeakingEQ 0 dB @ 20 Hz Q 0.707, disabled (:63-76). */
int  ohs_eq_create(int device, size_t num_bands, float fs, ohs_eq **out);
void ohs_eq_destroy(ohs_eq *eq);
/* update_band_coeffs(band_idx, sample_rate, &BandConfig)  parametric_eq.rs:144-164.
 * band >= num_bands is ignored (returns OHS_OK).  State (s1,s2) is kept (:112).
 * Coefficients by ohs_biquad_coeffs: convenience, NOT within 1e-6 of the reference unless that restatement agrees
 * with the `biquad` crate bit for bit (unpinned) -- see ohs_biquad_coeffs; the parity path is ohs_eq_set_band_coeffs. */
int  ohs_eq_update_band(ohs_eq *eq, size_t band, float fs, int filter_type, float fc, float q,
                        float gain_db, int enabled);
/* The parity path: the host supplies the coefficients (the `biquad` crate's, in the reference's shim), so the bits
 * are the reference's own; the kernel is bit-exact given them. */
int  ohs_eq_set_band_coeffs(ohs_eq *eq, size_t band, const float coeffs[5], int enabled);
int  ohs_eq_get_band_coeffs(const ohs_eq *eq, size_t band, float coeffs[5], int *enabled);
/* process_block(&mut [f32], &mut [f32]) in place  parametric_eq.rs:166-179.  Same products and sums
 * in the same order, each rounded separately: bit-identical to the reference's DF2T recurrence for
 * finite samples, except that a -0.0 may come out as +0.0; a non-finite INPUT sample makes the output
 * non-finite from up to 16 samples before it -- up to 64 in long launches on few chains, where a chain has a wave of
 * its own and 48 samples travel through its pass-on lanes at a time -- (the reference: from that sample on).  DESIGN.md 4.5,
 * LABNOTES.md. */
int  ohs_eq_process_block(ohs_eq *eq, float *left, float *right, size_t n);
/* Opt-in exactness for the two special-value corners above (default off): with enable != 0 the EQ runs its
 * "conveyor" kernel, in which samples move between lanes by register moves only -- a -0.0 stays -0.0 and a
 * non-finite input sample affects the output from that sample on, exactly as in the reference -- at ~1.4x the
 * time of the default kernel.  Audio hosts do not produce such samples; offline jobs that must be bit-exact on
 * arbitrary data switch it on. */
int  ohs_eq_set_exact_specials(ohs_eq *eq, int enable);
int  ohs_eq_set_flush_denormals(ohs_eq *eq, int mode);              /* see ohs_engine_set_flush_denormals */
/* reset_all_bands_state  parametric_eq.rs:181-188 (also Plugin::reset, lib.rs:1152-1154). */
int  ohs_eq_reset(ohs_eq *eq);
/* calculate_frequency_response(fs, &[f32]) -> Vec<f32>  parametric_eq.rs:190-209
 * (|H| of the enabled LEFT bands). */
int  ohs_eq_frequency_response(const ohs_eq *eq, float fs, const float *freqs, size_t n, float *out);

/* ---- BiquadFilter  (src/dsp/parametric_eq.rs:46-123) as a type of its own ------------------------------------------
 * The reference exports it (`pub struct`), uses it only inside StereoParametricEQ, and tests it directly
 * (parametric_eq.rs:218-238).  One DF2T section on one mono chain; the same kernels as the EQ, so the same bits. */
typedef struct ohs_biquad ohs_biquad;
/* BiquadFilter::new(initial_sample_rate)  :63-76: PeakingEQ 0 dB @ 20 Hz Q 0.707, DISABLED */
int  ohs_biquad_create(int device, float initial_sample_rate, ohs_biquad **out);
int  ohs_biquad_clone(const ohs_biquad *src, ohs_biquad **out);     /* impl Clone  :52-60: coefficients, state, enabled */
void ohs_biquad_destroy(ohs_biquad *f);
int  ohs_biquad_reset_state(ohs_biquad *f);                         /* :78-80 */
int  ohs_biquad_set_enabled(ohs_biquad *f, int enabled);            /* :82-84 */
/* update_coeffs(filter_type, sample_rate, center_freq, q, gain_db)  :86-114; state kept (:112).  Coefficients by
 * ohs_biquad_coeffs: the convenience path (see there); ohs_biquad_set_coeffs takes the `biquad` crate's own. */
int  ohs_biquad_update_coeffs(ohs_biquad *f, int filter_type, float sample_rate, float center_freq, float q, float gain_db);
int  ohs_biquad_set_coeffs(ohs_biquad *f, const float coeffs[5]);
/* process_sample(input) -> output  :116-122 over n consecutive samples, in place (n == 1 is the reference's call);
 * disabled => the samples come back bit for bit (:118-120). */
int  ohs_biquad_process(ohs_biquad *f, float *samples, size_t n);

/* ---- AutoEQ profile ingestion  (src/autoeq_parser.rs:21-70) -- SURVEY section 8 row f3 ---------- */
/* BandSetting  autoeq_parser.rs:34-41 */
typedef struct ohs_band_setting {
    int   enabled;       /* always 1 for parsed rows (autoeq_parser.rs:60) */
    int   filter_type;   /* OHS_FILTER_PEAK / LOWSHELF / HIGHSHELF ("PK" / "LS" / "HS", :43-50) */
    float frequency;     /* column "Fc"   */
    float q;             /* column "Q"    */
    float gain;          /* column "Gain" */
} ohs_band_setting;
/* parse_autoeq_csv(path) -> Vec<BandSetting>  autoeq_parser.rs:52-70: CSV with a header row naming
 * the columns Filter-Type, Fc, Q, Gain (any order, extra columns ignored, RFC-4180 quoting, empty
 * lines skipped -- the `csv` crate's defaults).  Writes at most `capacity` rows to `out`, the number
 * of rows in the file to `*count`.  OHS_ERR_IO if the file cannot be opened, OHS_ERR_PARSE for a
 * missing column, a field that is not an f32, a row of the wrong width or a filter type other than
 * PK / LS / HS ("Unsupported filter type: X", :48). */
int  ohs_autoeq_parse_csv(const char *path, ohs_band_setting *out, size_t capacity, size_t *count);
/* the same on an in-memory buffer */
int  ohs_autoeq_parse_csv_text(const char *text, size_t len, ohs_band_setting *out, size_t capacity,
                               size_t *count);

/* ---- SOFA HRIR source  (src/sofa/loader.rs:79-199 + system libmysofa) -- SURVEY section 8 row f1 -- */
/* A minimal reader for AES69 SimpleFreeFieldHRIR files (HDF5 subset, see csrc/sofa_reader.cpp).
 * ohs_sofa_open returns the file's own measurements untouched; libmysofa's conditioning (resampling inside
 * open, loudness normalisation, neighbour interpolation) is opt-in through ohs_sofa_open_ex.  Angles are DEGREES, AES69 convention: azimuth
 * counter-clockwise from the front (positive = left), elevation up; radius in metres. */
typedef struct ohs_sofa ohs_sofa;
/* MySofa::open(path, ..)  loader.rs:79-130.  OHS_ERR_IO when the file cannot be opened
 * (SofaError::FileOpen, tested at loader.rs:244-257), OHS_ERR_PARSE when it is not a readable SOFA file. */
int  ohs_sofa_open(const char *path, ohs_sofa **out);
/* mysofa_open as the reference calls it (loader.rs:83-90) also CONDITIONS the data; ohs_sofa_open_ex does the same,
 * step by step opt-in.  PARITY UNPINNED: libmysofa is not vendored under the reference and not available here;
 * the steps restate its published algorithm (csrc/sofa_conditioning.cpp cites the libmysofa functions).
 *   OHS_SOFA_RESAMPLE     every IR resampled to target_fs (Data.Delay scaled), as mysofa_resample -- with this
 *                         build's Kaiser-sinc resampler, not libmysofa's speex resampler
 *   OHS_SOFA_LOUDNESS     mysofa_loudness: all IRs scaled so that the frontal measurement's two ears have energy 2
 *   OHS_SOFA_INTERPOLATE  ohs_sofa_get_hrtf_irs = mysofa_getfilter_float: query projected onto the measured radius
 *                         range, inverse-distance interpolation of the nearest measurement with its neighbours in
 *                         azimuth / elevation / radius, delays likewise */
enum { OHS_SOFA_RESAMPLE = 1, OHS_SOFA_LOUDNESS = 2, OHS_SOFA_INTERPOLATE = 4, OHS_SOFA_MYSOFA_DEFAULTS = 7 };
int  ohs_sofa_open_ex(const char *path, unsigned flags, float target_fs, ohs_sofa **out);
int  ohs_sofa_loudness_factor(const ohs_sofa *s, float *factor);    /* what OHS_SOFA_LOUDNESS multiplied by (1 if off) */
/* the six neighbours mysofa_neighborhood would hold for measurement m: +az, -az, +el, -el, +r, -r (-1 = none) */
int  ohs_sofa_neighbors(const ohs_sofa *s, size_t m, int out[6]);
void ohs_sofa_close(ohs_sofa *s);                                   /* Drop -> mysofa_close  loader.rs:64-73 */
/* M measurements x R receivers x N taps, Data.SamplingRate (hrtf.N / DataSamplingRate, loader.rs:102-118) */
int  ohs_sofa_info(const ohs_sofa *s, size_t *M, size_t *R, size_t *N, float *fs);
int  ohs_sofa_position(const ohs_sofa *s, size_t m, float az_el_r[3]);
/* MySofa::spherical_to_cartesian / cartesian_to_spherical  loader.rs:203-229 ([az deg, el deg, r m] <-> xyz) */
int  ohs_sofa_spherical_to_cartesian(const float az_el_r[3], float xyz[3]);
int  ohs_sofa_cartesian_to_spherical(const float xyz[3], float az_el_r[3]);
/* nearest measurement in Cartesian space (what mysofa_lookup's kd-tree returns) */
int  ohs_sofa_nearest(const ohs_sofa *s, float az_deg, float el_deg, float radius_m, size_t *index);
int  ohs_sofa_get_ir(const ohs_sofa *s, size_t m, size_t receiver, float *out, size_t capacity);
/* MySofa::get_hrtf_irs(az, el, r) -> (left_ir, right_ir)  loader.rs:136-199; also returns the
 * per-ear Data.Delay values the reference leaves as a TODO (loader.rs:193-196).  Degrees, not the
 * radians the reference hands to mysofa_s2c (loader.rs:158-165). */
int  ohs_sofa_get_hrtf_irs(const ohs_sofa *s, float az_deg, float el_deg, float radius_m, float *left,
                           float *right, size_t capacity, float *delay_left, float *delay_right);
/* Kaiser-windowed-sinc resampling of one IR, e.g. 44.1 -> 48 kHz (libmysofa does this inside
 * mysofa_open, loader.rs:83-90).  out == NULL: only *out_len = ceil(n * fs_out / fs_in) is written. */
int  ohs_sofa_resample_ir(const float *in, size_t n, float fs_in, float fs_out, float *out, size_t capacity,
                          size_t *out_len);

/* ---- speaker angles -> the four paths (the wiring the reference plans but does not have) --------------------------
 * The plugin's CLAP parameters az_l / el_l / az_r / el_r (src/lib.rs:120-128) are smoothed every callback and then
 * DISCARDED (src/lib.rs:1170-1173); MySofa::get_hrtf_irs (src/sofa/loader.rs:136-199) and ConvolutionEngine::set_ir are
 * never connected (github_issues/sofa_implement_logic_select_extract_hrirs.md:5 asks for exactly that: nearest
 * measurement per speaker, the four HRIRs LSL / LSR / RSL / RSR handed to the engine).  ohs_*_set_speakers is that
 * connection: two ohs_sofa_get_hrtf_irs, each response resampled to `fs` (fs <= 0, or the file's own rate: untouched),
 * left speaker -> (OHS_PATH_LSL, OHS_PATH_LSR), right speaker -> (OHS_PATH_RSL, OHS_PATH_RSR), and set_ir ONLY for the
 * paths whose response differs from what the previous ohs_*_set_speakers call on this handle loaded (a set_ir from
 * anywhere else makes that path count as changed) -- with nearest-neighbour lookup a small move of a speaker usually
 * changes nothing, and a path that is not re-loaded keeps its history and overlap (convolution.rs:135-138 resets them).
 * Angles are the PLUGIN's: degrees, azimuth positive to the RIGHT (src/ui/speaker_visualizer.rs:51-54; defaults
 * az_l = -30, az_r = +30), elevation up; the SOFA query uses the AES69 azimuth -az.  *changed_mask (may be NULL): bit p
 * set when path p was re-loaded.  Not for the audio thread: a set_ir allocates and waits for the device. */
/* the four responses [Lsl, Lsr, Rsl, Rsr] ohs_*_set_speakers would load (no device needed): lens[p] = their lengths,
 * out[p] (may be NULL, as may out) receives at most `capacity` samples of response p */
int  ohs_sofa_speaker_irs(const ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r, float radius_m, float fs,
                          float *const out[4], size_t capacity, size_t lens[4]);
int  ohs_engine_set_speakers(ohs_engine *e, const ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r,
                             float radius_m, float fs, unsigned *changed_mask);

/* ---- Plugin::process DSP chain  (src/lib.rs:1169-1207) ----------------------- */
/* In place on left/right: bypass => untouched; [eq_enable => EQ in place] ->
 * convolution -> every sample *= gain.  One H2D + one D2H per call; the caller
 * refreshes EQ coefficients beforehand (lib.rs:1180-1193). eq may be NULL when
 * eq_enable == 0. */
int  ohs_chain_process(ohs_engine *e, ohs_eq *eq, int bypass, int eq_enable, float gain,
                       float *left, float *right, size_t n);

/* ---- offline many-stream batch mode (north_star; no reference counterpart) ---- */
/* n_streams independent stereo streams share one HRIR set and one EQ table; all
 * per-stream state (EQ s1/s2, spectrum history, overlap) is device resident.
 * Audio buffers are DEVICE pointers (hipMalloc / torch.Tensor.data_ptr()),
 * planar: sample (s, ch, i) at base[s*stream_stride + ch*channel_stride + i]. */
typedef struct ohs_batch ohs_batch;

int  ohs_batch_create(int device, size_t n_streams, size_t num_bands, ohs_batch **out);
void ohs_batch_destroy(ohs_batch *b);
/* set_ir semantics of convolution.rs:111-139 applied to every stream. `ir` is a HOST pointer. */
int  ohs_batch_set_ir(ohs_batch *b, int path, const float *ir, size_t len);
/* speaker angles -> the four shared impulse responses, see ohs_engine_set_speakers */
int  ohs_batch_set_speakers(ohs_batch *b, const ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r,
                            float radius_m, float fs, unsigned *changed_mask);
int  ohs_batch_set_eq_band_coeffs(ohs_batch *b, size_t band, const float coeffs[5], int enabled);
/* convenience (coefficients by ohs_biquad_coeffs: not a parity path, see there) */
int  ohs_batch_update_eq_band(ohs_batch *b, size_t band, float fs, int filter_type, float fc,
                              float q, float gain_db, int enabled);
/* Per-stream EQ tables.  In the reference every plugin instance owns its bands (parametric_eq.rs:125-129, lib.rs:316-324:
 * StereoParametricEQ::update_band_coeffs acts on that instance alone); an offline job of N listeners with N AutoEQ profiles
 * (ohs_autoeq_*) is N instances.  ohs_batch_set_stream_eq_band_coeffs gives stream `stream` (0 .. n_streams - 1) its own
 * band `band`: update_band_coeffs semantics for that stream -- coefficients and enable flag replaced, its state kept
 * (parametric_eq.rs:144-164), band >= num_bands ignored.  The first such call gives every stream a copy of the shared
 * table; from then on ohs_batch_set_eq_band_coeffs / _update_eq_band set the band of EVERY stream, and
 * ohs_batch_share_eq_table goes back to the one shared table (the values the shared calls have set).  All streams run in one
 * launch per 12 enabled bands of the busiest stream (each row of the EQ kernel reads its own stream's table when its wave
 * starts; streams may enable different bands and different NUMBERS of bands; in a later pass the streams with fewer bands
 * hand their samples on) as long as the exact-specials mode is off; in that mode every stream is a launch sequence of its
 * own -- correct for any table, slow for many streams.  The tables are
 * uploaded by the first processing call after an edit, which waits for the work queued on its stream (not for the audio
 * thread; bit-exact against one StereoParametricEQ per stream either way). */
int  ohs_batch_set_stream_eq_band_coeffs(ohs_batch *b, size_t stream, size_t band, const float coeffs[5], int enabled);
/* convenience (coefficients by ohs_biquad_coeffs: not a parity path, see there) */
int  ohs_batch_update_stream_eq_band(ohs_batch *b, size_t stream, size_t band, float fs, int filter_type, float fc,
                                     float q, float gain_db, int enabled);
int  ohs_batch_share_eq_table(ohs_batch *b);
/* eq_enable / output gain of lib.rs:1179,1202 */
int  ohs_batch_set_eq_enabled(ohs_batch *b, int eq_enable);
int  ohs_batch_set_eq_exact_specials(ohs_batch *b, int enable);    /* see ohs_eq_set_exact_specials */
int  ohs_batch_set_flush_denormals(ohs_batch *b, int mode);         /* see ohs_engine_set_flush_denormals (EQ and convolution) */
int  ohs_batch_set_gain(ohs_batch *b, float gain);
/* Which kernel plan serves the convolution of a batch call.
 *   1  block 512 / FFT 1024 -- the reference's own blocking (convolution.rs:22-23,120-132), for every impulse response
 *      length.  One partition (taps <= 512): output bits do not depend on where a signal is cut into calls, and are the ones
 *      tests/golden/p1_output_sha256.txt records.
 *   2  the large-transform plans, wherever their preconditions hold (else plan 1 serves the call):
 *        every path ONE partition (BASELINE configs[2] / [4]): hop 1536 / FFT 2048, overlap-save -- a quarter less transform
 *          and product work per frame, 11-12 % less time where the convolution is what the step waits for.  Needs even
 *          strides and 8-byte aligned buffers, calls of 3 blocks or more.
 *        longer responses (taps > 512; BASELINE configs[3]): block 2048 / FFT 4096, overlap-save -- a quarter of the
 *          reference blocking's products and history reads per frame (16 384 taps: 8 partitions instead of the 32 of
 *          convolution.rs:120-132), product and inverse transform in one kernel.  Needs strides that are multiples of 4
 *          floats and 16-byte aligned buffers.  Out-of-place calls of 128 blocks or more on responses of at most 16 384 taps
 *          (with more than 8 192 taps: on 32 streams or more): block 8192 / FFT 16384 with both transforms and the product
 *          in ONE kernel -- two partitions at most, the second one's product carried in registers to the next block, no
 *          spectrum ever written to memory (BASELINE configs[3]: 0.116 ms per step against 0.135 for block 2048 and 0.27 for
 *          the reference blocking).
 *      Equal to plan 1 up to f32 rounding (2-4e-7 relative RMS against the oracle, like every FFT path), not bit for bit,
 *      and the rounding depends on where the calls cut the signal.  The plans hand the per-stream state over to each other:
 *      any sequence of calls under any sequence of plans is one continuous signal.
 *   0  the library's choice (default).  One partition: plan 2 for calls of 48 blocks or more on 512 streams or more -- on
 *      256 or more when the call is out of place and the EQ is off --, plan 1 otherwise (at BASELINE configs[2] the
 *      convolution hides under the EQ either way, and the block-512 bits stay as recorded).  Longer responses: plan 2 for every
 *      call -- block 2048's blocks sit on the stream's absolute grid, so calls of any length continue one another, and the
 *      block-8192 kernel starts from the same state at rest (the streams' last input frames).
 * The single-stream engine is always served by the reference blocking. */
int  ohs_batch_set_conv_plan(ohs_batch *b, int plan);
/* What served the handle's most recent convolution launch (of the last time chunk, when a call is cut into several): the
 * kernel family and the number of independent ranges a stream's frames were cut into for it (one-partition plans: time
 * chunks per stream; block 2048: output tiles per stream of the last segment; block 8192: runs of blocks per stream; time-parallel block 512: blocks of the
 * last segment; sequential kernel: 1).  A benchmark or a test asks instead of restating the library's rule. */
enum ohs_conv_kernel {
    OHS_CONV_KERNEL_NONE = 0,           /* no convolution launch yet */
    OHS_CONV_KERNEL_BLOCK512_P1 = 1,    /* k_conv_p1: one partition, block 512 / FFT 1024 */
    OHS_CONV_KERNEL_HOP1536_P1 = 2,     /* k_conv_p1_os: one partition, hop 1536 / FFT 2048 */
    OHS_CONV_KERNEL_BLOCK512_TP = 3,    /* k_conv_tp_*: several partitions, block 512, time-parallel */
    OHS_CONV_KERNEL_BLOCK2048 = 4,      /* k_conv_lb_*: several partitions, block 2048 / FFT 4096 */
    OHS_CONV_KERNEL_SEQUENTIAL = 5,     /* k_conv_general: one wave per stream, blocks in sequence */
    OHS_CONV_KERNEL_BLOCK8192 = 6,      /* k_conv_xb: at most two partitions of 8192 taps, block 8192 / FFT 16384, one kernel */
    OHS_CONV_KERNEL_COUNT = 8           /* length of ohs_batch_conv_plan_counts' array (7 is not reported by this library) */
};
int  ohs_batch_last_conv_plan(const ohs_batch *b, int *kernel, int *ranges_per_stream);
/* convolution launch sequences served per kernel family (counts[OHS_CONV_KERNEL_*]) since the handle was created or since
 * the last call with reset != 0: a processing call that is cut into time chunks counts once per chunk, and the chunks of
 * one call need not be served alike (the short last chunk of an overlapped step usually is not) */
int  ohs_batch_conv_plan_counts(ohs_batch *b, uint64_t counts[8] /* OHS_CONV_KERNEL_COUNT */, int reset);
/* zero every stream's EQ state, history and overlap (IRs and EQ table kept).  Also the way back from a failed handle:
 * a HIP failure in the middle of a processing call leaves the per-stream state advanced for some time chunks only, so
 * the call joins its internal stream, returns OHS_ERR_HIP, and every later processing call on the handle returns
 * OHS_ERR_HIP as well until ohs_batch_reset. */
int  ohs_batch_reset(ohs_batch *b);
/* Process n_blocks * 512 frames of every stream: [EQ] -> convolution -> gain.
 * Asynchronous on `hip_stream` (a hipStream_t, NULL = default stream).
 * d_in may equal d_out. Strides are in floats. */
int  ohs_batch_process(ohs_batch *b, const float *d_in, float *d_out, size_t n_blocks,
                       size_t stream_stride, size_t channel_stride, void *hip_stream);
/* The same, but `hip_stream` is NOT made to wait for the last time chunk's convolution (it runs on an
 * internal stream underneath the EQ): d_out is complete on `hip_stream` only after ohs_batch_join (a
 * stream-side wait, asynchronous) or ohs_batch_sync.  Back-to-back deferred calls with the same buffers
 * pipeline: the next call's EQ starts while the previous call's last convolution finishes; every other
 * use of the handle joins by itself first.  Results are identical to ohs_batch_process. */
int  ohs_batch_process_deferred(ohs_batch *b, const float *d_in, float *d_out, size_t n_blocks,
                                size_t stream_stride, size_t channel_stride, void *hip_stream);
int  ohs_batch_join(ohs_batch *b, void *hip_stream);
/* The same processing with HOST buffers (same planar layout and strides, host pointers): the frames are cut
 * into time chunks of `chunk_blocks` blocks (0 = library default, ~32 MiB per direction) and copy-in of
 * chunk i + 1, the kernels of chunk i and copy-out of chunk i - 1 run concurrently on three internal
 * streams.  Blocking: h_out is complete on return.  h_in may equal h_out.  Pinned (page-locked) host memory
 * is needed for the copies to overlap at link speed; pageable memory works but serialises.  Results are
 * bit-identical to the sequence of ohs_batch_process calls with the same chunk sizes on device buffers.
 * The call runs on internal streams: work queued earlier on this handle through ohs_batch_process /
 * _process_deferred on a stream of the caller's must have completed (ohs_batch_sync) -- it advances the same
 * per-stream state; a pending deferred call is joined by the library itself. */
int  ohs_batch_process_host(ohs_batch *b, const float *h_in, float *h_out, size_t n_blocks,
                            size_t stream_stride, size_t channel_stride, size_t chunk_blocks);
/* waits (host side) for everything queued on `hip_stream` and for a pending deferred call */
int  ohs_batch_sync(ohs_batch *b, void *hip_stream);
/* Per-kernel timing for the roofline report: when enabled, every ohs_batch_process call brackets
 * its kernels with hipEvents on the caller's stream.  ohs_batch_profile_read waits for them,
 * returns the accumulated milliseconds per kernel, the number of ohs_batch_process calls and the
 * number of timed EQ / convolution launches since the last read, and clears the accumulators.
 * (With the EQ || convolution overlap a call issues 6 EQ launches and 6 convolution launches.) */
int  ohs_batch_set_profiling(ohs_batch *b, int enable);
int  ohs_batch_profile_read(ohs_batch *b, double *eq_ms, double *conv_ms, uint64_t *n_calls,
                            uint64_t *eq_launches, uint64_t *conv_launches);
/* With profiling on, wave 0 of every ring-form EQ launch also leaves two counters of its start and end behind (s_memrealtime,
 * 100 MHz, and s_memtime, the shader clock): the clock the serial recurrence REALLY ran at -- what turns ns per sample into
 * cycles per sample (the nominal 2.4 GHz is not what a loaded chip runs at) -- and that wave's lifetime, of
 * the most recent such launch.  Waits for the device. */
int  ohs_batch_profile_eq_clock(ohs_batch *b, double *shader_ghz, double *wave_us);
/* HBM bytes of one ohs_batch_process call of n_blocks by SURVEY.md 8d's per-block model (state persisted
 * between 512-frame blocks: 25 112 B per stream-block at P = 1) -- the yardstick of the whole path */
int  ohs_batch_algorithmic_bytes(const ohs_batch *b, size_t n_blocks, uint64_t *bytes);
/* DESIGN bytes per kernel family: what the EQ / convolution kernels are built to move through HBM for such a
 * call (P = 1: 16 B per frame + chunk-boundary tails + state; it writes no history and keeps overlaps in
 * registers) -- the numerator of a kernel's own HBM fraction, next to rocprofv3's measured bytes */
int  ohs_batch_kernel_bytes(const ohs_batch *b, size_t n_blocks, uint64_t *eq_bytes, uint64_t *conv_bytes);

/* ---- the batch mode over the GPUs of one node (north_star: 2048 streams over 8 MI355X) ----- */
/* One process, n_devices GPUs.  The job's streams_total independent stereo streams are cut into contiguous,
 * balanced id ranges (the first streams_total % n_devices devices get one more -- what every rank of bench.py
 * computes for itself), one ohs_batch per device.  The data path has NO collective: all mutable state is per stream
 * (convolution.rs:36-42,76-79; parametric_eq.rs:125-129).  The only exchange is the shared tables: they are uploaded
 * to the first device and reach the others by ONE ncclBroadcast over xGMI (RCCL, loaded from librccl.so when the
 * first node batch is created -- hosts that stay on one GPU never load it; communicators from ncclCommInitAll).
 * `devices` lists the HIP device indices (NULL = 0 .. n_devices - 1).  Same threading rule as every handle. */
typedef struct ohs_node_batch ohs_node_batch;
int  ohs_node_batch_create(int n_devices, const int *devices, size_t streams_total, size_t num_bands,
                           ohs_node_batch **out);
void ohs_node_batch_destroy(ohs_node_batch *nb);
/* the shard of device slot `slot` (0 .. n_devices - 1): HIP device index, first stream id, number of streams */
int  ohs_node_batch_shard(const ohs_node_batch *nb, int slot, int *device, size_t *first_stream, size_t *n_streams);
/* the per-device handle (owned by the node batch), for callers that drive a device themselves */
int  ohs_node_batch_device_batch(ohs_node_batch *nb, int slot, ohs_batch **out);
/* All shared tables at once: four impulse responses (HOST pointers, lens[p] == 0 => that path is muted) and the EQ
 * table (coeffs: num_bands x {b0,b1,b2,a1,a2}, enabled: num_bands ints; both may be NULL = leave the EQ table as
 * it is).  Packed, copied to the first device, ONE ncclBroadcast, unpacked on every device: set_ir semantics of
 * convolution.rs:111-139 for all four paths of every stream, ohs_batch_set_eq_band_coeffs for every band. */
int  ohs_node_batch_set_tables(ohs_node_batch *nb, const float *const irs[4], const size_t lens[4],
                               const float *eq_coeffs, const int *eq_enabled);
/* one path in mid-stream (one broadcast of that impulse response) */
int  ohs_node_batch_set_ir(ohs_node_batch *nb, int path, const float *ir, size_t len);
/* speaker angles -> the four shared impulse responses (ohs_engine_set_speakers), one broadcast per path that changed */
int  ohs_node_batch_set_speakers(ohs_node_batch *nb, const ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r,
                                 float radius_m, float fs, unsigned *changed_mask);
/* host-side settings, applied to every device's handle (no device traffic: kernel arguments) */
int  ohs_node_batch_set_eq_band_coeffs(ohs_node_batch *nb, size_t band, const float coeffs[5], int enabled);
/* one stream's own band (ohs_batch_set_stream_eq_band_coeffs): `stream` is the job's stream id, 0 .. streams_total - 1; it
 * is host-side state of the one device handle that holds the stream -- no broadcast */
int  ohs_node_batch_set_stream_eq_band_coeffs(ohs_node_batch *nb, size_t stream, size_t band, const float coeffs[5],
                                              int enabled);
int  ohs_node_batch_share_eq_table(ohs_node_batch *nb);
int  ohs_node_batch_set_eq_enabled(ohs_node_batch *nb, int eq_enable);
int  ohs_node_batch_set_gain(ohs_node_batch *nb, float gain);
int  ohs_node_batch_set_conv_plan(ohs_node_batch *nb, int plan);       /* see ohs_batch_set_conv_plan */
int  ohs_node_batch_reset(ohs_node_batch *nb);
/* ohs_batch_process_host for the whole job: h_in / h_out hold ALL streams_total streams (same planar layout and
 * strides); every device runs the three-stage host pipeline on its shard from a thread of its own.  Blocking.
 * Pinned host memory must be visible to every device (hipHostMallocPortable).  Streams do not interact; the results are
 * bit-identical to ohs_batch_process_host on one device with the same chunk_blocks WHEN BOTH SIDES RUN THE SAME PLAN
 * (ohs_batch_set_conv_plan / ohs_node_batch_set_conv_plan): the library's own choice (plan 0) follows the number of
 * streams PER DEVICE and the chunk length -- an unsharded batch of 512 streams or more with chunks of 48 blocks or more
 * takes the hop-1536 plan where its 8-way shards of 256 streams or fewer keep the block-512 plan -- and the two plans
 * agree to f32 rounding (<= 1e-6 relative RMS against the reference either way), not bit for bit.  Pin the plan where bits
 * must match across shardings (see ohs_node_batch_process for impulse responses longer than one partition). */
int  ohs_node_batch_process_host(ohs_node_batch *nb, const float *h_in, float *h_out, size_t n_blocks,
                                 size_t stream_stride, size_t channel_stride, size_t chunk_blocks);
/* THE DATA PATH OF THE NODE: ohs_batch_process on every device at once, audio resident in device memory.
 * d_in[slot] / d_out[slot] (slot = 0 .. n_devices - 1) are DEVICE pointers on that slot's device holding that shard's
 * n_streams(slot) streams, planar, same strides (in floats) on every device; d_in[slot] may equal d_out[slot].  Every
 * device's launches are issued by a host thread of its own (created with the node batch) on a stream of its own, so
 * the devices' queues fill side by side; nothing crosses PCIe or xGMI -- all mutable state is per stream
 * (convolution.rs:36-42,76-79; parametric_eq.rs:125-129).  Returns when every device has QUEUED its work: the outputs
 * are complete after ohs_node_batch_sync.  Calls queue up in order; the set_* / reset calls of the node batch wait for
 * what is queued.  Results are bit-identical to ohs_batch_process on one device holding the same streams when every
 * path has one partition (taps <= 512) and both sides run the same plan of ohs_batch_set_conv_plan -- the library's own
 * choice depends on the number of streams per device, so pin the plan (ohs_node_batch_set_conv_plan) where bits must
 * match across different shardings; either plan computes a block / a hop on its own.  With longer impulse responses the
 * time-parallel kernels cut a call into segments of max(8, min(16384 / streams on the device, 1024)) blocks and round
 * differently across a segment or call boundary (stored overlap, added in time) than inside one (overlap folded into
 * the spectrum): a shard and the unsharded batch agree bit for bit only while they cut alike, otherwise within the
 * parity bar (<= 1e-6 relative RMS against the reference either way).
 * Failure: if ANY slot fails the call returns that slot's status ("device slot k: ...") and the node batch refuses
 * every later ohs_node_batch_process / _process_host with OHS_ERR_HIP until ohs_node_batch_reset (the other shards have
 * advanced: the node's streams are out of step).  Likewise after a table call that failed half-way (devices hold
 * different tables) until a complete ohs_node_batch_set_tables -- four paths and the EQ -- has succeeded. */
int  ohs_node_batch_process(ohs_node_batch *nb, const float *const d_in[], float *const d_out[], size_t n_blocks,
                            size_t stream_stride, size_t channel_stride);
/* waits (host side) for everything ohs_node_batch_process has queued, on every device.  On a node batch that has failed
 * (above) it still waits for every device, then returns OHS_ERR_HIP: what the buffers hold is not a result. */
int  ohs_node_batch_sync(ohs_node_batch *nb);
/* the stream (hipStream_t) ohs_node_batch_process queues slot's work on: a caller that produces its audio on the device
 * orders its own kernels against it with events */
int  ohs_node_batch_stream(ohs_node_batch *nb, int slot, void **hip_stream);
/* device time of what is queued between the two calls, per device slot (HIP events on the slots' streams; _end waits):
 * ms_per_slot has n_devices entries.  The node's rate is the job's frames / the largest entry.  Both refuse a failed node
 * batch (OHS_ERR_HIP): a time over a half-advanced step is not a step time. */
int  ohs_node_batch_timer_begin(ohs_node_batch *nb);
int  ohs_node_batch_timer_end(ohs_node_batch *nb, float *ms_per_slot);
/* what RCCL reports for the communicators in use: ncclCommCount of slot 0's, and 1 if librccl.so is loaded */
int  ohs_node_batch_rccl_info(const ohs_node_batch *nb, int *comm_size, int *rccl_loaded);

#ifdef __cplusplus
}
#endif
#endif /* OHS_HIP_H */
