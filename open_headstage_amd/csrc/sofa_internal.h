// sofa_internal.h -- the opened SOFA file as sofa_reader.cpp and sofa_conditioning.cpp share it (not public).
#pragma once
#include "../../include/ohs_hip.h"

#include <vector>

struct ohs_sofa {
    size_t M = 0, R = 0, N = 0;
    float fs = 0.f;
    std::vector<float> ir;          // [M][R][N]
    std::vector<float> pos_sph;     // [M][3] azimuth deg, elevation deg, radius m
    std::vector<float> pos_xyz;     // [M][3]
    std::vector<float> delay;       // [M][R] (or broadcast of [1][R]), samples
    // ohs_sofa_open_ex (libmysofa-style conditioning)
    unsigned flags = 0;
    float loudness_factor = 1.0f;   // what OHS_SOFA_LOUDNESS multiplied every IR by
    double radius_min = 0.0, radius_max = 0.0;
    bool has_phi = false, has_theta = false, has_radius = false;     // the grid varies in that coordinate
    mutable std::vector<int> neighbors;      // [M][6] (+phi, -phi, +theta, -theta, +r, -r), -2 = not computed yet, -1 = none
};

extern "C" int ohsint_sofa_interpolated(const ohs_sofa *s, float az_deg, float el_deg, float radius_m, float *left,
                                              float *right, size_t capacity, float *delay_left, float *delay_right);
