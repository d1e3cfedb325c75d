// sofa_conditioning.cpp -- what libmysofa's mysofa_open / mysofa_getfilter_float do to a SOFA file beyond reading it
// (the reference reaches them through src/sofa/loader.rs:83-90 and :174-185).
//
// PARITY UNPINNED: libmysofa is a system library of the reference, not vendored under it and not on this
// machine; the steps below restate its published algorithm (easy.c mysofa_open_default, loudness.c,
// neighbors.c, interpolate.c, lookup.c of libmysofa 1.x) from knowledge of that library:
//   resample   every IR to the target rate, samples scaled by fs_in / fs_out, Data.Delay by fs_out / fs_in
//              (libmysofa: speex resampler, quality 10 -- here this build's Kaiser-windowed sinc, so the
//              resampled taps differ in the low-order bits and in the transition band)
//   loudness   all IRs times sqrt(2 / E), E = energy of both ears of the "frontal" measurement, which
//              libmysofa defines as the one minimising azimuth + elevation in degrees (ties: larger radius)
//   lookup     query projected onto the measured radius range, nearest measurement in Cartesian space
//   neighbours of a measurement: the first DIFFERENT nearest measurement met when stepping +/- 0.5 degrees in
//              azimuth, +/- 0.5 degrees in elevation (up to 45) and +/- 0.01 m in radius
//   interpolate inverse-distance weights over the nearest measurement and, per coordinate, the closer of its
//              two neighbours (neither if they are equally far); delays likewise
// Opt-in (ohs_sofa_open_ex); ohs_sofa_open keeps returning the file's own measurements untouched.
#include "sofa_internal.h"

#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>

extern "C" void ohsint_set_error(const char *msg);
namespace {
int cfail(int code, const std::string &m) { ohsint_set_error(m.c_str()); return code; }
const float kPiF = 3.14159265358979323846f;

// libmysofa's mysofa_s2c / mysofa_c2s (degrees), in f32 as there
void s2c_f(float v[3])
{
    const float phi = v[0] * (kPiF / 180.0f), theta = v[1] * (kPiF / 180.0f), r = v[2];
    const float x = std::cos(theta) * r;
    v[2] = std::sin(theta) * r;
    v[0] = std::cos(phi) * x;
    v[1] = std::sin(phi) * x;
}
void c2s_f(float v[3])
{
    const float x = v[0], y = v[1], z = v[2];
    const float r = std::sqrt(x * x + y * y + z * z);
    const float theta = std::atan2(z, std::sqrt(x * x + y * y)), phi = std::atan2(y, x);
    v[0] = std::fmod(phi * (180.0f / kPiF) + 360.0f, 360.0f);
    v[1] = theta * (180.0f / kPiF);
    v[2] = r;
}
bool fequals(float a, float b) { return std::fabs(a - b) < 0.00001f; }
float dist3(const float *a, const float *b)
{
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return std::sqrt(dx * dx + dy * dy + dz * dz);
}

// mysofa_lookup: project onto [radius_min, radius_max] (in place, as libmysofa does), nearest in Cartesian space
int lookup(const ohs_sofa *s, float c[3])
{
    const float r = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    if (r > (float)s->radius_max && r > 0) { const float f = (float)s->radius_max / r; c[0] *= f; c[1] *= f; c[2] *= f; }
    else if (r < (float)s->radius_min && r > 0) { const float f = (float)s->radius_min / r; c[0] *= f; c[1] *= f; c[2] *= f; }
    float best = 3.4e38f;
    int bi = 0;
    for (size_t m = 0; m < s->M; ++m) {
        const float *p = &s->pos_xyz[m * 3];
        const float dx = p[0] - c[0], dy = p[1] - c[1], dz = p[2] - c[2];
        const float d = dx * dx + dy * dy + dz * dz;
        if (d < best) { best = d; bi = (int)m; }
    }
    return bi;
}

// neighbors.c for ONE measurement (computed on first use, cached)
const int *neighborhood(const ohs_sofa *s, int i)
{
    int *nb = &s->neighbors[(size_t)i * 6];
    if (nb[0] != -2) return nb;
    for (int k = 0; k < 6; ++k) nb[k] = -1;
    float origin[3] = {s->pos_xyz[(size_t)i * 3], s->pos_xyz[(size_t)i * 3 + 1], s->pos_xyz[(size_t)i * 3 + 2]};
    c2s_f(origin);
    const float angle_step = 0.5f, radius_step = 0.01f;
    auto probe = [&](float dphi, float dtheta, float dr) -> int {
        float t[3] = {origin[0] + dphi, origin[1] + dtheta, origin[2] + dr};
        s2c_f(t);
        return lookup(s, t);
    };
    if (s->has_phi) {
        for (float phi = angle_step; phi <= 45.0f; phi += angle_step) { const int j = probe(phi, 0, 0); if (j != i) { nb[0] = j; break; } }
        for (float phi = -angle_step; phi >= -45.0f; phi -= angle_step) { const int j = probe(phi, 0, 0); if (j != i) { nb[1] = j; break; } }
    }
    if (s->has_theta) {
        for (float th = angle_step; th <= 45.0f; th += angle_step) { const int j = probe(0, th, 0); if (j != i) { nb[2] = j; break; } }
        for (float th = -angle_step; th >= -45.0f; th -= angle_step) { const int j = probe(0, th, 0); if (j != i) { nb[3] = j; break; } }
    }
    if (s->has_radius) {
        for (float r = radius_step; origin[2] + r <= (float)s->radius_max + radius_step; r += radius_step) {
            const int j = probe(0, 0, r); if (j != i) { nb[4] = j; break; }
        }
        for (float r = -radius_step; origin[2] + r >= (float)s->radius_min - radius_step; r -= radius_step) {
            const int j = probe(0, 0, r); if (j != i) { nb[5] = j; break; }
        }
    }
    return nb;
}
}  // namespace

extern "C" {

int ohs_sofa_open_ex(const char *path, unsigned flags, float target_fs, ohs_sofa **out)
{
    if (!path || !out) return cfail(OHS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    if (flags & ~(unsigned)OHS_SOFA_MYSOFA_DEFAULTS) return cfail(OHS_ERR_INVALID_ARG, "unknown flag");
    if ((flags & OHS_SOFA_RESAMPLE) && !(target_fs > 0)) return cfail(OHS_ERR_INVALID_ARG, "target sample rate must be > 0");
    ohs_sofa *raw = nullptr;
    const int rc = ohs_sofa_open(path, &raw);
    if (rc) return rc;
    std::unique_ptr<ohs_sofa, void (*)(ohs_sofa *)> s(raw, ohs_sofa_close);
    try {
        // mysofa_resample
        if ((flags & OHS_SOFA_RESAMPLE) && s->fs > 0 && !fequals(target_fs, s->fs)) {
            const float factor = target_fs / s->fs;
            size_t newN = 0;
            if (ohs_sofa_resample_ir(s->ir.data(), s->N, s->fs, target_fs, nullptr, 0, &newN)) return OHS_ERR_INVALID_ARG;
            if (newN == 0 || newN > (1u << 20)) return cfail(OHS_ERR_PARSE, "resampled filter length out of range");
            std::vector<float> nir(s->M * s->R * newN);
            for (size_t i = 0; i < s->M * s->R; ++i) {
                size_t got = 0;
                ohs_sofa_resample_ir(&s->ir[i * s->N], s->N, s->fs, target_fs, &nir[i * newN], newN, &got);
            }
            s->ir.swap(nir);
            s->N = newN;
            for (float &d : s->delay) d *= factor;
            s->fs = target_fs;
        }
        // radius / angle ranges (lookup.c) -- also needed by the loudness step's frontal search
        s->radius_min = 1e30; s->radius_max = 0;
        float phi_min = 1e30f, phi_max = -1e30f, th_min = 1e30f, th_max = -1e30f;
        for (size_t m = 0; m < s->M; ++m) {
            float c[3] = {s->pos_xyz[m * 3], s->pos_xyz[m * 3 + 1], s->pos_xyz[m * 3 + 2]};
            c2s_f(c);
            phi_min = std::fmin(phi_min, c[0]); phi_max = std::fmax(phi_max, c[0]);
            th_min = std::fmin(th_min, c[1]); th_max = std::fmax(th_max, c[1]);
            s->radius_min = std::fmin(s->radius_min, (double)c[2]); s->radius_max = std::fmax(s->radius_max, (double)c[2]);
        }
        s->has_phi = (phi_max - phi_min) > 1.2e-38f;
        s->has_theta = (th_max - th_min) > 1.2e-38f;
        s->has_radius = (float)(s->radius_max - s->radius_min) > 1.2e-38f;
        // mysofa_loudness
        if ((flags & OHS_SOFA_LOUDNESS) && s->M > 0) {
            float mn = 3.4e38f;
            int radius = 0;             // an int in libmysofa too
            size_t index = 0;
            for (size_t m = 0; m < s->M; ++m) {
                const float a = s->pos_sph[m * 3], e = s->pos_sph[m * 3 + 1], r = s->pos_sph[m * 3 + 2];
                if (mn > a + e) { mn = a + e; radius = (int)r; index = m; }
                else if (mn == a + e && (float)radius < r) { radius = (int)r; index = m; }
            }
            float energy = 0.f;
            const float *f = &s->ir[index * s->R * s->N];
            for (size_t i = 0; i < s->R * s->N; ++i) energy += f[i] * f[i];
            if (energy > 0) {
                const float factor = std::sqrt(2.0f / energy);
                if (!fequals(factor, 1.0f)) {
                    for (float &v : s->ir) v *= factor;
                    s->loudness_factor = factor;
                }
            }
        }
        if (flags & OHS_SOFA_INTERPOLATE) s->neighbors.assign(s->M * 6, -2);
        s->flags = flags;
    } catch (const std::bad_alloc &) {
        return cfail(OHS_ERR_ALLOC, "out of memory while conditioning the SOFA data");
    } catch (const std::exception &e) {
        return cfail(OHS_ERR_PARSE, std::string("SOFA conditioning failed: ") + e.what());
    }
    *out = s.release();
    return OHS_OK;
}

int ohs_sofa_loudness_factor(const ohs_sofa *s, float *factor)
{
    if (!s || !factor) return cfail(OHS_ERR_INVALID_ARG, "NULL argument");
    *factor = s->loudness_factor;
    return OHS_OK;
}

int ohs_sofa_neighbors(const ohs_sofa *s, size_t m, int out[6])
{
    if (!s || !out || m >= s->M) return cfail(OHS_ERR_INVALID_ARG, "bad argument");
    if (!(s->flags & OHS_SOFA_INTERPOLATE)) return cfail(OHS_ERR_INVALID_ARG, "file was not opened with OHS_SOFA_INTERPOLATE");
    try {
        const int *nb = neighborhood(s, (int)m);
        for (int k = 0; k < 6; ++k) out[k] = nb[k];
    } catch (...) { return cfail(OHS_ERR_ALLOC, "out of memory"); }
    return OHS_OK;
}

// mysofa_getfilter_float: lookup + mysofa_interpolate
int ohsint_sofa_interpolated(const ohs_sofa *s, float az_deg, float el_deg, float radius_m, float *left, float *right,
                                   size_t capacity, float *delay_left, float *delay_right)
{
    float c[3] = {az_deg, el_deg, radius_m};
    s2c_f(c);
    const int nearest = lookup(s, c);                 // c is now the projected query, as in libmysofa
    const size_t size = s->N * s->R, n = capacity < s->N ? capacity : s->N;
    const float *pos = s->pos_xyz.data();
    const float d = dist3(c, pos + (size_t)nearest * 3);
    auto copy_out = [&](const float *fir, float dl, float dr) {
        std::memcpy(left, fir, n * sizeof(float));
        std::memcpy(right, fir + s->N, n * sizeof(float));
        if (delay_left) *delay_left = dl;
        if (delay_right) *delay_right = dr;
    };
    if (fequals(d, 0.0f)) {
        copy_out(&s->ir[(size_t)nearest * size], s->delay[(size_t)nearest * s->R], s->delay[(size_t)nearest * s->R + 1]);
        return OHS_OK;
    }
    const int *nb;
    try { nb = neighborhood(s, nearest); } catch (...) { return cfail(OHS_ERR_ALLOC, "out of memory"); }
    int use[6] = {0, 0, 0, 0, 0, 0};
    float d6[6] = {1, 1, 1, 1, 1, 1};
    for (int pair = 0; pair < 3; ++pair) {
        const int a = 2 * pair, b = a + 1;
        if (nb[a] >= 0 && nb[b] >= 0) {
            d6[a] = dist3(c, pos + (size_t)nb[a] * 3);
            d6[b] = dist3(c, pos + (size_t)nb[b] * 3);
            if (!fequals(d6[a], d6[b])) { if (d6[a] < d6[b]) use[a] = 1; else use[b] = 1; }
        } else if (nb[a] >= 0) {
            use[a] = 1; d6[a] = dist3(c, pos + (size_t)nb[a] * 3);
        } else if (nb[b] >= 0) {
            use[b] = 1; d6[b] = dist3(c, pos + (size_t)nb[b] * 3);
        }
    }
    std::vector<float> fir;
    try { fir.assign(size, 0.0f); } catch (...) { return cfail(OHS_ERR_ALLOC, "out of memory"); }
    float weight = 1.0f / d;
    const float *src = &s->ir[(size_t)nearest * size];
    for (size_t i = 0; i < size; ++i) fir[i] = src[i] * weight;
    float dl = s->delay[(size_t)nearest * s->R] * weight, dr = s->delay[(size_t)nearest * s->R + 1] * weight;
    for (int k = 0; k < 6; ++k) {
        if (!use[k]) continue;
        const float w = 1.0f / d6[k];
        const float *q = &s->ir[(size_t)nb[k] * size];
        for (size_t i = 0; i < size; ++i) fir[i] += q[i] * w;
        weight += w;
        dl += s->delay[(size_t)nb[k] * s->R] * w;
        dr += s->delay[(size_t)nb[k] * s->R + 1] * w;
    }
    weight = 1.0f / weight;
    for (size_t i = 0; i < size; ++i) fir[i] *= weight;
    copy_out(fir.data(), dl * weight, dr * weight);
    return OHS_OK;
}

}  // extern "C"
