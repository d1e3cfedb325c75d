// eq_ring2_body.hpp -- the ring-form DF2T cascade (<= 12 bands) on 25 pinned VGPRs: device code shared by
// eq_kernels.hip (k_eq_ring) and conv_kernels.hip (the resident real-time kernel runs the plugin's whole chain).
// Every file that includes this header MUST be built with -ffp-contract=off: the recurrence rounds each product
// and sum separately, as the reference does (src/dsp/parametric_eq.rs:116-122).
//
// Same arithmetic, lane roles and issue order as round 2's form (experiments/eq_ring_v1.hpp, experiments build only); see
// eq_ring_body.hpp and tools/gen_eq_ring2_asm.py for the ring, the two-lane split of a band and the I/O port.  What
// is new here is the register plan: round 2's asm bodies named v100..v113 as temporaries, which made k_eq_ring a
// 114-VGPR kernel -- and k_conv_p1 takes 4 x 120 of a SIMD's 512 registers, so an EQ wave and a convolution
// workgroup could never share a CU: the two kernels partitioned the chip (<= 1024 streams) or serialised (2048).
// Now every value of the steady state lives in a pinned register v0..v23 (v24 in the two NOLOAD forms; v25 the second
// store register of the interior groups, whose memory instructions go out two chunks at a time), the eight
// input registers are reloaded in place, addresses are SGPR base + one 32-bit VGPR offset, and the interior groups
// loop inside ONE asm statement.  The kernel as a whole stays within 32 VGPRs.
//
// Software pipeline across blocks.  A step is   O T A N (this step)  |  port + P + M (the NEXT step's)
// and every block -- asm or the gated C++ form below -- starts behind the `|` of its first step and ends behind the
// `|` of its last, so that blocks chain without a prologue of their own: RingRegs carries X, the store register,
// (s1, s2), (b0x, b1x) and b2x between them in the registers the asm pins.
#pragma once
#include "kernels.h"
#include "eq_ring_body.hpp"     // v2f, dpp helpers, RingLane, ring_port (shared with the round-2 form)

namespace ohs {

#include "eq_ring2_asm.inc"

struct RingRegs {
    float X, st;        // v2, v3
    v2f u;              // v[4:5]   (b0, b1) * X of the pre lane
    float b2x;          // v7       b2 * X(row_ror:1) of the post lane
    v2f s;              // v[10:11] (s1, s2)
};

// port + P + M in front of step (16 q + s): what every block leaves behind for its successor
template <int B>
__device__ __forceinline__ void ring2_pre(RingRegs &r, const RingLane &c, float xin)
{
    ring_port<B>(r.st, r.X, xin);
}
__device__ __forceinline__ void ring2_pm(RingRegs &r, const RingLane &c)
{
    r.u = (v2f){c.pb0 * r.X, c.pb1 * r.X};
    r.b2x = c.b2 * dpp_mov<kRowRor1>(r.X, r.X);
}

// One chunk = 16 steps in the gated form (bands waking up at the start of a launch, the ragged end).
//   q      chunk index;  xq = this chunk's 16 inputs (banks 1..3 are injected here, bank 0 was by the predecessor)
//   xnext  chunk q + 1's inputs (its bank 0 is injected behind step 15)
//   dst0 + lane_off   this lane's sample slot of chunk 0 (wave-uniform base + 32-bit byte offset)
// 32-bit sample indices: the ring form only runs where eq_ring2_addressable() holds
__device__ __forceinline__ float ring2_ld(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void ring2_st(float *base, unsigned byte_off, float v)
{
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
__device__ __forceinline__ void ring2_chunk(int q, int n, int j, bool band, const RingLane &c, RingRegs &r,
                                            float xq, float xnext, float *dst0, unsigned lane_off)
{
    const int u0 = q * 16 - j;                      // sample index this post lane filters at step 0
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float out = dpp_mov<kRowRor1>(r.u.x, r.u.x) + r.s.x;          // O
        const float t2 = dpp_mov<kRowRor1>(r.u.y, r.u.y) + r.s.y;           // T
        const float n1 = t2 - c.a1 * out;                                   // A, N
        const float n2 = r.b2x - c.a2 * out;
        const float x = dpp_mov<kRowRor1>(r.X, r.X);
        const bool upd = band && (unsigned)(u0 + s) < (unsigned)n;
        r.s.x = upd ? n1 : r.s.x;
        r.s.y = upd ? n2 : r.s.y;
        r.X = upd ? out : x;
        // the next step's port
        if (s == 3) ring2_pre<1>(r, c, xq);
        if (s == 7) ring2_pre<2>(r, c, xq);
        if (s == 11) {
            ring2_pre<3>(r, c, xq);
            const int yi = (q - 1) * 16 + (j ^ 3);
            if ((unsigned)yi < (unsigned)n) ring2_st(dst0, lane_off + (unsigned)(q - 1) * 64u, r.st);
        }
        if (s == 15) ring2_pre<0>(r, c, xnext);
        ring2_pm(r, c);
    }
}

// pinned-register operand lists of the asm blocks
#define EQ_RING2_STATE [X] "+{v2}"(r.X), [st] "+{v3}"(r.st), [u] "+{v[4:5]}"(r.u), [b2x] "+{v7}"(r.b2x), [s] "+{v[10:11]}"(r.s)
#define EQ_RING2_CONSTS [b2] "{v1}"(c.b2), [pb01] "{v[12:13]}"(pb01), [a12] "{v[14:15]}"(a12)
#define EQ_RING2_CLOBBERS "v6", "v8", "v9", "memory"

// The launch's full groups, n_full >= 1 of them, in ONE statement: chunks 1..7 of group 0 (chunk 0 ran in the gated
// form; the block requests chunks 2..7 itself, so that the gated code runs with two input registers alive, not
// eight), n_full - 2 interior groups, the last full group.  x1 = chunk 1's inputs, xnext = chunk 0 of the group
// behind the last full one.
__device__ __forceinline__ void eq_ring2_full_groups(const RingLane &c, RingRegs &r, float x1, float xnext, unsigned lane_off,
                                                     const float *src, float *dst, int n_full)
{
    const v2f pb01 = {c.pb0, c.pb1}, a12 = {c.a1, c.a2};
    unsigned k512 = 512;
    // (wave-uniform by construction; readfirstlane tells hipcc so -- it had put the selects into VGPRs)
    unsigned only = (unsigned)__builtin_amdgcn_readfirstlane(n_full == 1 ? 1 : 0);
    unsigned interior_groups = (unsigned)__builtin_amdgcn_readfirstlane(n_full >= 2 ? n_full - 2 : 0);
    asm volatile(EQ_RING2_MAIN
                 : EQ_RING2_STATE, [x1] "+{v17}"(x1), [voff] "+{v0}"(lane_off), [cnt] "+s"(interior_groups)
                 : EQ_RING2_CONSTS, [xn] "{v24}"(xnext), [src] "s"(src), [dst] "s"(dst), [k512] "s"(k512), [only] "s"(only)
                 : EQ_RING2_CLOBBERS, "v16", "v18", "v19", "v20", "v21", "v22", "v23", "v25", "v26", "v27", "scc");
}

// the strides the ring form's 32-bit lane offsets can address (host side, launch_eq_pass)
__host__ __device__ __forceinline__ bool eq_ring2_addressable(long long stream_stride, long long ch_stride, long long n)
{
    return stream_stride >= 0 && ch_stride >= 0 && (stream_stride + ch_stride + n + 1024) * 4 < (1ll << 32);
}

// One WAVE of the ring-form EQ: chains 4 * wave_id .. + 3 (rows beyond n_chains shadow a live row), the whole of
// n samples, state read at entry and written back at exit.  No barrier, no LDS: callable from any kernel.  The
// caller must have no vector-memory operation outstanding (the asm blocks count vmcnt by hand).
// PER_STREAM: every stream owns its bands (parametric_eq.rs:125-129: each plugin instance does) -- the constants, the
// state slots and the NUMBER of enabled bands of a row come from stabs[stream of the row] (kernels.h: EqStreamTable)
// instead of the launch's one table; a row with fewer bands simply has more pass-on lanes, the asm blocks do not know
// the difference.  The tables are read where the shared form reads its kernel argument: in the head, once more for the
// ragged end and the state write-back (no per-lane value of the compiler's may cross the asm blocks: 32 VGPRs).
template <bool PER_STREAM>
__device__ __forceinline__ void eq_ring2_wave_t(const float *in, float *out, long long stream_stride, long long ch_stride,
                                                long long n, int n_chains, int nb_shared, const EqPassTable &tab,
                                                const EqStreamTable *__restrict__ stabs, float *__restrict__ state,
                                                long long wave_id)
{
    constexpr int G = 8;            // chunks per group (128 samples per chain)
    if (wave_id * 4 >= n_chains) return;
    // The kernel has 32 VGPRs and the asm blocks pin 26 of them, so only ONE per-lane value of the compiler's
    // crosses the blocks: the lane id.  Each phase (head, ragged end, state write-back) derives its own copies of
    // the band index, the sample offset ... from a laundered lane id, which keeps hipcc from hoisting the common
    // subexpressions of the phases above the blocks and holding them in registers it does not have.
    struct Ids {
        int j, jp, row;
        bool band;              // a post lane: holds band j - 1's state
        unsigned lane_off;      // byte offset of this lane's slot of chunk 0 from the wave's base; + 64 per chunk
    };
    const int rows_here = n_chains - (int)wave_id * 4;
    const int live_rows = rows_here < 4 ? rows_here : 4;
    int shadow_mask = __builtin_amdgcn_readfirstlane(live_rows == 2 ? 1 : 0);
    asm volatile("" : "+s"(shadow_mask));       // an opaque SGPR: left to itself hipcc keeps it in a VGPR across the blocks
    auto stab_of = [&](int row) -> const EqStreamTable * { return stabs + (wave_id * 2 + (row >> 1)); };
    auto ids_of = [&](int lane_) -> Ids {
        Ids d;
        d.j = lane_ & 15;
        d.jp = d.j ^ 3;         // sample offset within a chunk held by this lane of an I/O register
        const int r_ = lane_ >> 4;
        // dead rows shadow a live row of the same wave (lock-step: same loads, same stores, no state write-back)
        d.row = r_ < live_rows ? r_ : (r_ & shadow_mask);       // = r_ % live_rows for live_rows in 1..4
        int nb = nb_shared;
        if constexpr (PER_STREAM) nb = stab_of(d.row)->nb;
        d.band = d.j >= 1 && d.j <= nb;
        const unsigned row_elems = (unsigned)((d.row >> 1) * stream_stride + (d.row & 1) * ch_stride);
        d.lane_off = (row_elems + (unsigned)d.jp) * 4u;
        return d;
    };
    auto state_offset = [&](const Ids &d) -> unsigned {
        int slot;
        if constexpr (PER_STREAM) slot = stab_of(d.row)->slot[d.band ? d.j - 1 : 0];
        else slot = tab.slot[d.band ? d.j - 1 : 0];
        return (unsigned)((d.row * kEqStateSlots + slot) * 2) * 4u;
    };
    auto launder = [](int v) { asm volatile("" : "+v"(v)); return v; };
    const int lane = threadIdx.x & 63;
    // wave-uniform bases (row 0 of the wave: the caller passes a uniform wave_id) + per-lane 32-bit byte offsets
    const float *src0 = in + (wave_id * 2) * stream_stride;
    float *dst0 = out + (wave_id * 2) * stream_stride;
    float *state0 = state + wave_id * (4 * kEqStateSlots * 2);

    // y[i] is captured during chunk i/16 + 1
    const int n32 = (int)n;
    const int n_chunks = (n32 - 1) / 16 + 2;
    const int n_full = n32 / (16 * G);                  // groups whose 128 samples all exist
    auto load_chunk = [&](const Ids &d, int q) -> float {       // chunk q's 16 inputs, zeros beyond n
        return q * 16 + d.jp < n32 ? ring2_ld(src0, d.lane_off + (unsigned)q * 64u) : 0.0f;
    };

    RingLane c;
    RingRegs r;
    float xq;
    int q = 0;
    {   // ---- head: constants, state, the launch's full groups
        const Ids d = ids_of(lane);
        const int jb = d.band ? d.j - 1 : 0;
        if constexpr (PER_STREAM) {
            const EqStreamTable *T = stab_of(d.row);
            const bool pre = d.j < T->nb;
            const int jp_ = pre ? d.j : 0;
            c.pb0 = pre ? T->b0[jp_] : 1.0f;
            c.pb1 = pre ? T->b1[jp_] : 0.0f;
            c.b2 = d.band ? T->b2[jb] : 0.0f;
            c.a1 = d.band ? T->a1[jb] : 0.0f;
            c.a2 = d.band ? T->a2[jb] : 0.0f;
        } else {
            const bool pre = d.j < nb_shared;
            c.pb0 = pre ? tab.b0[d.j] : 1.0f;
            c.pb1 = pre ? tab.b1[d.j] : 0.0f;
            c.b2 = d.band ? tab.b2[jb] : 0.0f;
            c.a1 = d.band ? tab.a1[jb] : 0.0f;
            c.a2 = d.band ? tab.a2[jb] : 0.0f;
        }
        const unsigned state_off = state_offset(d);
        r.s.x = d.band ? ring2_ld(state0, state_off) : 0.0f;
        r.s.y = d.band ? ring2_ld(state0, state_off + 4u) : 0.0f;
        r.X = 0.0f;
        r.st = 0.0f;
        // Chunk 0 of the ragged end (the first group that is not full): the last asm block injects it on its way
        // out.  Requested first, so that it is long there when that block starts (in place this is safe: nothing
        // stores to those samples before the ragged end runs).
        xq = load_chunk(d, n_full * G);
        if (n_full >= 1) {
            const float x0 = ring2_ld(src0, d.lane_off), x1 = ring2_ld(src0, d.lane_off + 64u);
            // pipeline prologue: step 0's port + P + M
            ring2_pre<0>(r, c, x0);
            ring2_pm(r, c);
            // group 0: chunk 0 gated (band j wakes up at step j + 1); the rest of the full groups as asm
            ring2_chunk(0, n32, d.j, d.band, c, r, x0, x1, dst0, d.lane_off);
            eq_ring2_full_groups(c, r, x1, xq, d.lane_off, src0, dst0, n_full);
            q = n_full * G;
        } else {
            ring2_pre<0>(r, c, xq);
            ring2_pm(r, c);
        }
    }
    {   // ---- the ragged end: at most 8 + 1 chunks, one at a time
        const Ids d = ids_of(launder(lane));
        for (; q < n_chunks; ++q) {
            const float xnext = load_chunk(d, q + 1);
            ring2_chunk(q, n32, d.j, d.band, c, r, xq, xnext, dst0, d.lane_off);
            xq = xnext;
        }
    }
    {   // ---- state write-back (live rows only)
        const int lane_c = launder(lane);
        const Ids d = ids_of(lane_c);
        if ((lane_c >> 4) < live_rows && d.band) {
            const unsigned state_off = state_offset(d);
            ring2_st(state0, state_off, r.s.x);
            ring2_st(state0, state_off + 4u, r.s.y);
        }
    }
}

// the shared-table form: what k_eq_ring and the resident real-time kernel run
__device__ __forceinline__ void eq_ring2_wave(const float *in, float *out, long long stream_stride, long long ch_stride,
                                              long long n, int n_chains, int nb, const EqPassTable &tab,
                                              float *__restrict__ state, long long wave_id)
{
    eq_ring2_wave_t<false>(in, out, stream_stride, ch_stride, n, n_chains, nb, tab, nullptr, state, wave_id);
}

}  // namespace ohs
