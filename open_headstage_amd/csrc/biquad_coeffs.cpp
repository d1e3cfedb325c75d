// biquad_coeffs.cpp -- the coefficient formulas behind BiquadFilter::update_coeffs
// (src/dsp/parametric_eq.rs:86-114 -> biquad 0.4.2 Coefficients::<f32>::from_params).  Host-only.
#include "../../include/ohs_hip.h"
#include "host_internal.h"
#include "libm_musl.h"

#include <cmath>
#include <string>

namespace ohs_host {

namespace {
int fail(int code, const std::string &m) { ohsint_set_error(m.c_str()); return code; }
}  // namespace

// biquad 0.4.2 Coefficients::<f32>::from_params, restated (see oracle/ohs_oracle.h for the caveat
// that biquad/libm are not vendored under the reference); f32 arithmetic, left to right.  The crate's sin / cos /
// powf are the `libm` crate's (a port of musl's), not the host's: libm_musl.h restates them, so that the result does
// not depend on which glibc this process runs on.  sqrtf is correctly rounded everywhere.
int rbj(int type, float fs, float fc, float q, float gain_db, float out[5])
{
    if (2.0f * fc > fs) return fail(OHS_ERR_OUTSIDE_NYQUIST, "2*fc > fs (biquad::Errors::OutsideNyquist)");
    if (q < 0.0f) return fail(OHS_ERR_NEGATIVE_Q, "q < 0 (biquad::Errors::NegativeQ)");
    if (type < 0 || type > 7) return fail(OHS_ERR_INVALID_ARG, "filter_type must be 0..7");
    const float pi = 3.14159265358979323846264338327950288f;
    const float omega = 2.0f * pi * fc / fs;
    const float sn = ohs_musl_sinf(omega), cs = ohs_musl_cosf(omega);
    const float alpha = sn / (2.0f * q);
    float b0, b1, b2, a0, a1, a2;
    switch (type) {
    case OHS_FILTER_LOWPASS:
        b0 = (1.0f - cs) * 0.5f; b1 = 1.0f - cs; b2 = (1.0f - cs) * 0.5f;
        a0 = 1.0f + alpha; a1 = -2.0f * cs; a2 = 1.0f - alpha; break;
    case OHS_FILTER_HIGHPASS:
        b0 = (1.0f + cs) * 0.5f; b1 = -(1.0f + cs); b2 = (1.0f + cs) * 0.5f;
        a0 = 1.0f + alpha; a1 = -2.0f * cs; a2 = 1.0f - alpha; break;
    case OHS_FILTER_BANDPASS:
        b0 = sn / 2.0f; b1 = 0.0f; b2 = -(sn / 2.0f);
        a0 = 1.0f + alpha; a1 = -2.0f * cs; a2 = 1.0f - alpha; break;
    case OHS_FILTER_NOTCH:
        b0 = 1.0f; b1 = -2.0f * cs; b2 = 1.0f;
        a0 = 1.0f + alpha; a1 = -2.0f * cs; a2 = 1.0f - alpha; break;
    case OHS_FILTER_ALLPASS:
        b0 = 1.0f - alpha; b1 = -2.0f * cs; b2 = 1.0f + alpha;
        a0 = 1.0f + alpha; a1 = -2.0f * cs; a2 = 1.0f - alpha; break;
    case OHS_FILTER_LOWSHELF: {
        const float a = ohs_musl_powf(10.0f, gain_db / 40.0f), sa = sqrtf(a);
        b0 = a * ((a + 1.0f) - (a - 1.0f) * cs + 2.0f * alpha * sa);
        b1 = 2.0f * a * ((a - 1.0f) - (a + 1.0f) * cs);
        b2 = a * ((a + 1.0f) - (a - 1.0f) * cs - 2.0f * alpha * sa);
        a0 = (a + 1.0f) + (a - 1.0f) * cs + 2.0f * alpha * sa;
        a1 = -2.0f * ((a - 1.0f) + (a + 1.0f) * cs);
        a2 = (a + 1.0f) + (a - 1.0f) * cs - 2.0f * alpha * sa; break; }
    case OHS_FILTER_HIGHSHELF: {
        const float a = ohs_musl_powf(10.0f, gain_db / 40.0f), sa = sqrtf(a);
        b0 = a * ((a + 1.0f) + (a - 1.0f) * cs + 2.0f * alpha * sa);
        b1 = -2.0f * a * ((a - 1.0f) + (a + 1.0f) * cs);
        b2 = a * ((a + 1.0f) + (a - 1.0f) * cs - 2.0f * alpha * sa);
        a0 = (a + 1.0f) - (a - 1.0f) * cs + 2.0f * alpha * sa;
        a1 = 2.0f * ((a - 1.0f) - (a + 1.0f) * cs);
        a2 = (a + 1.0f) - (a - 1.0f) * cs - 2.0f * alpha * sa; break; }
    default: {   // Peak
        const float a = ohs_musl_powf(10.0f, gain_db / 40.0f);
        b0 = 1.0f + alpha * a; b1 = -2.0f * cs; b2 = 1.0f - alpha * a;
        a0 = 1.0f + alpha / a; a1 = -2.0f * cs; a2 = 1.0f - alpha / a; break; }
    }
    out[0] = b0 / a0; out[1] = b1 / a0; out[2] = b2 / a0; out[3] = a1 / a0; out[4] = a2 / a0;
    return OHS_OK;
}

}  // namespace ohs_host
