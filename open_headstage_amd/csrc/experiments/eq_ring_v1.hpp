// eq_ring_v1.hpp -- round 2's ring-form EQ body (114 VGPRs: its asm blocks name v100..v113 as temporaries), superseded
// by eq_ring2_body.hpp in round 3.  EXPERIMENTS BUILD ONLY (k_eq_ring_v1, tuning key eq_ring_v1): the A/B of
// LABNOTES.md.  The generated asm it includes is tools/gen_eq_ring_asm.py's output.
#pragma once
#ifndef OHS_EXPERIMENTS
#error "eq_ring_v1.hpp belongs to the experiments build"
#endif
#include "../eq_ring_body.hpp"

namespace ohs {

// one chunk = 16 steps, generic (gated) form; q = chunk index, xin = this chunk's 16 inputs
__device__ __forceinline__ void ring_chunk(long long q, long long n, int j, int nb, const RingLane &c,
                                           float &s1, float &s2, float &X, float xin, float *dst)
{
    float store_reg = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (s == 0) ring_port<0>(store_reg, X, xin);
        if (s == 4) ring_port<1>(store_reg, X, xin);
        if (s == 8) ring_port<2>(store_reg, X, xin);
        if (s == 12) {
            ring_port<3>(store_reg, X, xin);
            const long long yi = (q - 1) * 16 + (j ^ 3);
            if (yi >= 0 && yi < n) dst[yi] = store_reg;
        }
        const float u0 = c.pb0 * X, u1 = c.pb1 * X;
        const float x = dpp_mov<kRowRor1>(X, X);
        const float out = dpp_mov<kRowRor1>(u0, u0) + s1;
        const float t2 = dpp_mov<kRowRor1>(u1, u1) + s2;
        const float n1 = t2 - c.a1 * out;
        const float n2 = c.b2 * x - c.a2 * out;
        const long long u = q * 16 + s - j;            // sample index this post lane is filtering
        const bool upd = j >= 1 && j <= nb && u >= 0 && u < n;
        s1 = upd ? n1 : s1;
        s2 = upd ? n2 : s2;
        X = upd ? out : x;
    }
}

#include "eq_ring_v1_asm.inc"
#define EQ_RING_CLOBBERS "v100", "v102", "v103", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "memory"
__device__ __forceinline__ void eq_ring_group_fast(v2f pb01, float b2, v2f a12, float &s1, float &s2, float &o,
                                                   float *sp, const float *lp, const float (&x)[8], float (&nx)[8])
{
    asm volatile(EQ_RING_BODY
                 : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o), [n0] "=&v"(nx[0]), [n1] "=&v"(nx[1]), [n2] "=&v"(nx[2]),
                   [n3] "=&v"(nx[3]), [n4] "=&v"(nx[4]), [n5] "=&v"(nx[5]), [n6] "=&v"(nx[6]), [n7] "=&v"(nx[7])
                 : [pb01] "v"(pb01), [b2] "v"(b2), [a12] "v"(a12), [sp] "v"(sp), [lp] "v"(lp), [x0] "v"(x[0]),
                   [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]),
                   [x7] "v"(x[7])
                 : EQ_RING_CLOBBERS);
}

// The same group without the prefetch (the caller has issued bounds-checked loads of the next group itself):
// the last group of a launch whose 128 samples all exist.  HEAD = chunks 1..7 only, for the launch's first
// group, whose chunk 0 (bands waking up one per step) runs in the gated form.
template <bool HEAD>
__device__ __forceinline__ void eq_ring_group_noload(v2f pb01, float b2, v2f a12, float &s1, float &s2, float &o,
                                                     float *sp, const float (&x)[8])
{
    if constexpr (HEAD) {
        asm volatile(EQ_RING_BODY_HEAD
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o)
                     : [pb01] "v"(pb01), [b2] "v"(b2), [a12] "v"(a12), [sp] "v"(sp), [x1] "v"(x[1]), [x2] "v"(x[2]),
                       [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7])
                     : EQ_RING_CLOBBERS);
    } else {
        asm volatile(EQ_RING_BODY_NOLOAD
                     : [s1] "+v"(s1), [s2] "+v"(s2), [o] "+v"(o)
                     : [pb01] "v"(pb01), [b2] "v"(b2), [a12] "v"(a12), [sp] "v"(sp), [x0] "v"(x[0]), [x1] "v"(x[1]),
                       [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7])
                     : EQ_RING_CLOBBERS);
    }
}

// One WAVE of the ring-form EQ: chains 4 * wave_id .. + 3 (rows beyond n_chains shadow a live row), the whole
// of n samples, state read at entry and written back at exit.  No barrier, no LDS: callable from any kernel
// (k_eq_ring below; the resident real-time kernel in conv_kernels.hip, whose file is built -ffp-contract=off
// like this one).  The caller must have no vector-memory operation outstanding (the asm groups count vmcnt by
// hand).
__device__ __forceinline__ void eq_ring_wave(const float *in, float *out, long long stream_stride, long long ch_stride,
                                             long long n, int n_chains, int nb, const EqPassTable &tab,
                                             float *__restrict__ state, long long wave_id)
{
    constexpr int G = 8;            // chunks per group (128 samples per chain)
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    if (wave_id * 4 >= n_chains) return;
    const long long chain = wave_id * 4 + (lane >> 4);
    const bool live = chain < n_chains;
    // dead rows shadow a live row of the same wave (see k_eq_pass)
    const long long live_rows = n_chains - wave_id * 4 < 4 ? n_chains - wave_id * 4 : 4;
    const long long cc = live ? chain : wave_id * 4 + (lane >> 4) % live_rows;
    const float *src = in + (cc >> 1) * stream_stride + (cc & 1) * ch_stride;
    float *dst = out + (cc >> 1) * stream_stride + (cc & 1) * ch_stride;

    const bool pre = j < nb, post = j >= 1 && j <= nb;
    const int jb = post ? j - 1 : 0;
    RingLane c;
    c.pb0 = pre ? tab.b0[j] : 1.0f;
    c.pb1 = pre ? tab.b1[j] : 0.0f;
    c.b2 = post ? tab.b2[jb] : 0.0f;
    c.a1 = post ? tab.a1[jb] : 0.0f;
    c.a2 = post ? tab.a2[jb] : 0.0f;
    const int slot = tab.slot[jb];
    float s1 = post ? state[(cc * kEqStateSlots + slot) * 2 + 0] : 0.0f;
    float s2 = post ? state[(cc * kEqStateSlots + slot) * 2 + 1] : 0.0f;
    float X = 0.0f;
    const int jp = j ^ 3;           // sample offset within a chunk held by this lane of an I/O register

    // y[i] is captured during chunk i/16 + 1
    const long long n_chunks = (n - 1) / 16 + 2;
    const long long n_groups = (n_chunks + G - 1) / G;
    // groups [1, g_fast_end): every band lane's sample index valid, prefetch and stores in bounds
    const long long g_fast_end = n / (16 * G) - 1;      // (g + 2) * 128 <= n

    float cur[G], nxt[G];
#pragma unroll
    for (int e = 0; e < G; ++e) {
        const long long xi = (long long)e * 16 + jp;
        cur[e] = (xi < n) ? src[xi] : 0.0f;
    }
    long long g = 0;
    while (g < n_groups) {
        if (g >= 1 && g < g_fast_end) {
            // interior groups in a loop with no compiler-tracked memory operation (see k_eq_pass);
            // retire the tracked loads of `cur` here, in front of the loop, not on its back edge
#pragma unroll
            for (int e = 0; e < G; ++e) asm volatile("" : "+v"(cur[e]));
            for (; g < g_fast_end; ++g) {
                const float *lp = src + (g + 1) * (16 * G) + jp;
                float *sp = dst + (g * G - 1) * 16 + jp;
                float xs[8];
#pragma unroll
                for (int e = 0; e < G; ++e) { xs[e] = cur[e]; asm volatile("" : "+v"(xs[e])); }
                eq_ring_group_fast((v2f){c.pb0, c.pb1}, c.b2, (v2f){c.a1, c.a2}, s1, s2, X, sp, lp, xs, nxt);
                eq_wait_loads<8>(nxt);
#pragma unroll
                for (int e = 0; e < G; ++e) cur[e] = nxt[e];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }
#pragma unroll
        for (int e = 0; e < G; ++e) {
            const long long xi = ((g + 1) * G + e) * 16 + jp;
            nxt[e] = (xi < n) ? src[xi] : 0.0f;
        }
        if ((g + 1) * (16 * G) <= n) {
            // all 128 samples of this group exist, only the next group's do not (or this is group 0): asm
            // without its own prefetch; group 0 runs its first chunk gated (band j wakes up at step j + 1)
            float *sp = dst + (g * G - 1) * 16 + jp;
            float xs[8];
#pragma unroll
            for (int e = 0; e < G; ++e) { xs[e] = cur[e]; asm volatile("" : "+v"(xs[e])); }
            if (g == 0) {
                ring_chunk(0, n, j, nb, c, s1, s2, X, xs[0], dst);
                eq_ring_group_noload<true>((v2f){c.pb0, c.pb1}, c.b2, (v2f){c.a1, c.a2}, s1, s2, X, sp, xs);
            } else {
                eq_ring_group_noload<false>((v2f){c.pb0, c.pb1}, c.b2, (v2f){c.a1, c.a2}, s1, s2, X, sp, xs);
            }
        } else {
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const long long q = g * G + e;
                if (q < n_chunks) ring_chunk(q, n, j, nb, c, s1, s2, X, cur[e], dst);
            }
        }
#pragma unroll
        for (int e = 0; e < G; ++e) cur[e] = nxt[e];
        ++g;
    }
    if (live && post) {
        state[(cc * kEqStateSlots + slot) * 2 + 0] = s1;
        state[(cc * kEqStateSlots + slot) * 2 + 1] = s2;
    }
}

}  // namespace ohs
