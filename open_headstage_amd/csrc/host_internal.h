// host_internal.h -- host-side helpers shared between the translation units of libohs_hip.so (not part
// of the public header).
#pragma once

#include <stddef.h>

#include <vector>

extern "C" void ohsint_set_error(const char *msg);     // api_core.hip: feeds ohs_last_error()
struct ohs_batch;
struct ohs_sofa;
// api_batch.hip: ohs_batch_set_ir with the impulse response already in memory of the batch's device
extern "C" int ohsint_batch_set_ir_device(ohs_batch *b, int path, const float *d_ir, size_t len);

namespace ohs_host {

// biquad 0.4.2 Coefficients::<f32>::from_params restated; out = {b0, b1, b2, a1, a2} / a0.
// Returns an OHS_* status (OHS_ERR_OUTSIDE_NYQUIST, OHS_ERR_NEGATIVE_Q, OHS_ERR_INVALID_ARG).
int rbj(int type, float fs, float fc, float q, float gain_db, float out[5]);

// speakers.cpp: the four impulse responses [Lsl, Lsr, Rsl, Rsr] of two virtual speakers at the plugin's angles
// (degrees, azimuth positive to the right), optionally resampled to fs
int speaker_irs(const struct ::ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r, float radius_m, float fs,
                std::vector<float> out[4]);

}  // namespace ohs_host
