// eq_kernels.hip -- 10-band parametric EQ (DF2T biquad cascade) for gfx950.
//
// Computes exactly what src/dsp/parametric_eq.rs:116-122,166-179 computes
// (biquad 0.4.2 DirectForm2Transposed::run):
//     out = s1 + b0*x ;  s1 = (s2 + b1*x) - a1*out ;  s2 = b2*x - a2*out
// with every product and sum rounded separately -- this file is compiled with
// -ffp-contract=off and the result is BIT-EXACT against the CPU oracle.  An IIR
// is a serial recurrence in time, so the parallelism is (a) across chains
// (stream x channel) and (b) ACROSS BANDS, systolically:
//
//   a 16-lane DPP row is one chain; lane j of the row is band j and keeps that
//   band's (s1, s2, coefficients) in VGPRs for the whole launch.  Every step each
//   lane filters one sample and hands its output to lane j+1 with a single
//   `row_shr:1` DPP move, so band j works on sample t-j while band j+1 works on
//   sample t-j-1.  The arithmetic each band performs, and its order, are those
//   of the reference's sample-major / band-minor loop.
//
// Samples enter at lane 0 and leave at lane 15, 16 at a time: each row loads /
// stores 64 contiguous bytes per 16 steps.  More than 16 bands = several passes.
#include "kernels.h"

namespace ohs {

__device__ __forceinline__ float row_bcast(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

// lane j of each 16-lane row receives lane j-1's `v`; lane 0 of the row receives `feed`.
__device__ __forceinline__ float row_shift_in(float v, float feed)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(feed), __float_as_int(v),
                                                      0x111 /* row_shr:1 */, 0xf, 0xf, false));
}

template <bool GATED>
__device__ __forceinline__ void eq_chunk(long long t0, long long n, int j, int rowbase, bool en,
                                         float b0, float b1, float b2, float a1, float a2,
                                         float &s1, float &s2, float &prev, float xchunk, float &ychunk)
{
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        // lane 15's output of the previous step is y[t0 + k - 16]
        const float ydone = row_bcast(prev, rowbase + 15);
        ychunk = (j == k) ? ydone : ychunk;
        const float feed = row_bcast(xchunk, rowbase + k);
        const float x = row_shift_in(prev, feed);
        const float out = s1 + b0 * x;
        const float n1 = (s2 + b1 * x) - a1 * out;
        const float n2 = b2 * x - a2 * out;
        bool upd = en;
        if (GATED) {
            const long long u = t0 + k - j;        // sample index this lane is filtering
            upd = en && u >= 0 && u < n;
        }
        s1 = upd ? n1 : s1;
        s2 = upd ? n2 : s2;
        prev = upd ? out : x;                      // disabled band = identity (parametric_eq.rs:118-120)
    }
}

__global__ __launch_bounds__(64) void k_eq_pass(const float *in, float *out,
                                                long long stream_stride, long long ch_stride,
                                                long long n, int n_chains, const EqPassTable tab,
                                                float *__restrict__ state)
{
    const int lane = threadIdx.x;
    const int j = lane & 15;
    const int rowbase = lane & 48;
    const long long chain = (long long)blockIdx.x * 4 + (lane >> 4);
    const bool live = chain < n_chains;
    const long long cc = live ? chain : 0;
    const float *src = in + (cc >> 1) * stream_stride + (cc & 1) * ch_stride;
    float *dst = out + (cc >> 1) * stream_stride + (cc & 1) * ch_stride;

    const float b0 = tab.b0[j], b1 = tab.b1[j], b2 = tab.b2[j], a1 = tab.a1[j], a2 = tab.a2[j];
    const bool en = tab.en[j] != 0;
    float s1 = state[(cc * 16 + j) * 2 + 0];
    float s2 = state[(cc * 16 + j) * 2 + 1];
    float prev = 0.0f, ychunk = 0.0f;

    // steps t = 0 .. n+15; chunk q covers steps 16q .. 16q+15 and completes y[16(q-1) .. 16q-1]
    const long long n_chunks = (n + 16 + 15) / 16;
    for (long long q = 0; q < n_chunks; ++q) {
        const long long t0 = q * 16;
        const long long xi = t0 + j;
        const float xchunk = (live && xi < n) ? src[xi] : 0.0f;
        // a chunk needs gating when some lane's sample index t0+k-j can fall outside [0, n)
        if (t0 >= 16 && t0 + 16 <= n) eq_chunk<false>(t0, n, j, rowbase, en, b0, b1, b2, a1, a2, s1, s2, prev, xchunk, ychunk);
        else                          eq_chunk<true>(t0, n, j, rowbase, en, b0, b1, b2, a1, a2, s1, s2, prev, xchunk, ychunk);
        // after step 16q+15, ychunk (lane j) = y[16(q-1) + j]
        if (q >= 1) {
            const long long yi = t0 - 16 + j;
            if (live && yi < n) dst[yi] = ychunk;
        }
    }
    if (live) {
        state[(cc * 16 + j) * 2 + 0] = s1;
        state[(cc * 16 + j) * 2 + 1] = s2;
    }
}

hipError_t launch_eq_pass(const float *in, float *out, long long stream_stride, long long ch_stride,
                          long long n, int n_chains, const EqPassTable &tab, float *state, hipStream_t st)
{
    if (n <= 0 || n_chains <= 0) return hipSuccess;
    const int blocks = (n_chains + 3) / 4;
    hipLaunchKernelGGL(k_eq_pass, dim3(blocks), dim3(64), 0, st, in, out, stream_stride, ch_stride, n,
                       n_chains, tab, state);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// calculate_frequency_response (parametric_eq.rs:190-209), num-complex 0.4.6 op order
// ---------------------------------------------------------------------------------
__global__ void k_eq_freq_response(const float *__restrict__ coeffs, const int *__restrict__ en, int nb,
                                   float fs, const float *__restrict__ freqs, int n, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float pi = 3.14159265358979323846264338327950288f;
    float rr = 1.0f, ri = 0.0f;
    for (int b = 0; b < nb; ++b) {
        if (!en[b]) continue;
        const float b0 = coeffs[5 * b], b1 = coeffs[5 * b + 1], b2 = coeffs[5 * b + 2];
        const float a1 = coeffs[5 * b + 3], a2 = coeffs[5 * b + 4];
        const float omega = 2.0f * pi * freqs[i] / fs;
        const float zr = 1.0f * cosf(-omega), zi = 1.0f * sinf(-omega);      // from_polar(1, -omega)
        const float ns = zr * zr + zi * zi;
        const float ir = zr / ns, ii = -zi / ns;                               // z.powi(-1)
        const float i2r = ir * ir - ii * ii, i2i = ir * ii + ii * ir;          // z.powi(-2)
        const float nr = (b0 + b1 * ir) + b2 * i2r, ni = b1 * ii + b2 * i2i;
        const float dr = (1.0f + a1 * ir) + a2 * i2r, di = a1 * ii + a2 * i2i;
        const float dn = dr * dr + di * di;
        const float qr = (nr * dr + ni * di) / dn, qi = (ni * dr - nr * di) / dn;
        const float tr = rr * qr - ri * qi, ti = rr * qi + ri * qr;
        rr = tr; ri = ti;
    }
    out[i] = hypotf(rr, ri);
}

hipError_t launch_eq_freq_response(const float *d_coeffs, const int *d_en, int nb, float fs,
                                   const float *d_freqs, int n, float *d_out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_eq_freq_response, dim3((n + 63) / 64), dim3(64), 0, st, d_coeffs, d_en, nb, fs,
                       d_freqs, n, d_out);
    return hipGetLastError();
}

}  // namespace ohs
