// speakers.cpp -- the wiring the reference plans but does not have: the plugin's four speaker-angle parameters
// (CLAP ids az_l / el_l / az_r / el_r, src/lib.rs:120-128; their smoothed values are discarded at :1170-1173) ->
// MySofa::get_hrtf_irs (src/sofa/loader.rs:136-199) for each speaker -> four set_ir calls
// (github_issues/sofa_implement_logic_select_extract_hrirs.md:5: "select the nearest available HRTF measurement ...
// extract the four required HRIRs (LSL, LSR, RSL, RSR)").  Host-only C++ on top of the SOFA reader.
#include "../../include/ohs_hip.h"
#include "host_internal.h"

#include <cmath>
#include <string>
#include <vector>

namespace ohs_host {

// The left speaker's (left-ear, right-ear) responses become Lsl / Lsr, the right speaker's Rsl / Rsr.
// Angles are the PLUGIN's: degrees, azimuth positive to the RIGHT (the editor draws a speaker at x = sin(az),
// src/ui/speaker_visualizer.rs:51-54; defaults az_l = -30, az_r = +30, src/lib.rs:429-432), elevation up.  SOFA / AES69
// azimuth is counter-clockwise (positive = left), hence the sign.  fs > 0: every response resampled from the file's
// rate to fs when they differ by more than 1e-3 Hz (libmysofa does this inside mysofa_open, loader.rs:83-90); fs <= 0:
// the file's own samples.
int speaker_irs(const ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r, float radius_m, float fs,
                std::vector<float> out[4])
{
    size_t M = 0, R = 0, N = 0;
    float file_fs = 0.f;
    int rc = ohs_sofa_info(sofa, &M, &R, &N, &file_fs);
    if (rc) return rc;
    const float az[2] = {-az_l, -az_r}, el[2] = {el_l, el_r};
    for (int spk = 0; spk < 2; ++spk) {
        std::vector<float> ear[2] = {std::vector<float>(N), std::vector<float>(N)};
        float dl = 0.f, dr = 0.f;
        rc = ohs_sofa_get_hrtf_irs(sofa, az[spk], el[spk], radius_m, ear[0].data(), ear[1].data(), N, &dl, &dr);
        if (rc) return rc;
        for (int e = 0; e < 2; ++e) {
            std::vector<float> &dst = out[2 * spk + e];
            if (fs > 0.f && std::fabs(fs - file_fs) > 1e-3f) {
                size_t n_out = 0;
                rc = ohs_sofa_resample_ir(ear[e].data(), N, file_fs, fs, nullptr, 0, &n_out);
                if (rc) return rc;
                dst.assign(n_out, 0.0f);
                rc = ohs_sofa_resample_ir(ear[e].data(), N, file_fs, fs, dst.data(), dst.size(), &n_out);
                if (rc) return rc;
            } else {
                dst = ear[e];
            }
        }
    }
    return OHS_OK;
}

}  // namespace ohs_host

extern "C" {

// the four responses themselves, for hosts that want them without a device (and for the CPU test suite)
int ohs_sofa_speaker_irs(const ohs_sofa *sofa, float az_l, float el_l, float az_r, float el_r, float radius_m, float fs,
                         float *const out[4], size_t capacity, size_t lens[4])
{
    if (!sofa || !lens) { ohsint_set_error("NULL argument"); return OHS_ERR_INVALID_ARG; }
    std::vector<float> irs[4];
    const int rc = ohs_host::speaker_irs(sofa, az_l, el_l, az_r, el_r, radius_m, fs, irs);
    if (rc) return rc;
    for (int p = 0; p < 4; ++p) {
        lens[p] = irs[p].size();
        if (out && out[p])
            for (size_t i = 0; i < irs[p].size() && i < capacity; ++i) out[p][i] = irs[p][i];
    }
    return OHS_OK;
}

}  // extern "C"
