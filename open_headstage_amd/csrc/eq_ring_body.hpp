// eq_ring_body.hpp -- what the forms of the DF2T cascade share: DPP helpers, the hand-counted vmcnt wait, the ring's
// lane roles and its I/O port (the description of the ring form lives here; eq_ring2_body.hpp is the kernel body built
// on it, experiments/eq_ring_v1.hpp -- experiments build only -- round 2's).
// Every file that includes this header MUST be built with -ffp-contract=off: the recurrence rounds each
// product and sum separately, as the reference does (src/dsp/parametric_eq.rs:116-122).
#pragma once
#include "kernels.h"

namespace ohs {

typedef float v2f __attribute__((ext_vector_type(2)));

// DPP move: lanes whose source lane lies outside their 16-lane row keep `old`.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL,
                                                      0xf, 0xf, false));
}
constexpr int kRowShr1 = 0x111;   // lane j <- lane j-1
constexpr int kRowShl1 = 0x101;   // lane j <- lane j+1

// Untracked (inline-asm) global accesses for the interior of the stream.  hipcc cannot count
// vmcnt across the chunk loop and falls back to `s_waitcnt vmcnt(0)` in front of every chunk,
// which serialises each chunk behind the store it has just issued (+~500 cycles per 16 samples).
// Here the queue is counted by hand: per group of G = 8 chunks one wave issues 8 prefetch loads,
// (inside the asm group), then 8 stores, and `s_waitcnt vmcnt(8)` (all but the 8 youngest = the stores)
// retires the loads.
// wait until at most N vector-memory operations of this wave are outstanding; ties the loaded
// registers to the wait so that no use of them can be scheduled above it
template <int N>
__device__ __forceinline__ void eq_wait_loads(float (&v)[8])
{
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "n"(N)
                 : "memory");
}

// ---------------------------------------------------------------------------------
// Ring form (NB <= 12): 6.5 VALU per sample instead of 9
// ---------------------------------------------------------------------------------
// The conveyor register of k_eq_pass costs three DPP moves per step (hand-over, shift, insert).  Here
// the row itself is the conveyor: the 16 lanes form a ring (`row_ror:1`).  Register X: lane L holds the
// sample band L filters next; lanes that hold no band pass their sample on unchanged, so a finished
// sample leaving band NB-1 keeps travelling towards lane 15 and whatever sits in lane 15 enters band 0
// on the next step.  Band L's arithmetic is split over two lanes,
//     lane L   ("pre") : (b0x, b1x) = (b0, b1) * X
//     lane L+1 ("post"): out = s1 + b0x ;  s1 = (s2 + b1x) - a1*out ;  s2 = b2*x - a2*out
// where b0x, b1x and x arrive through the DPP operand (row_ror:1) of the instruction that consumes them:
// the output lands in lane L+1 of X, which is where band L+1 wants its input -- the hand-over costs no
// instruction.  (s1 + b0x is evaluated as b0x + s1: IEEE addition commutes bit for bit.)
// Bank 3 (lanes 12..15) is the I/O port: every 4 steps it holds 4 finished samples, which one
// bank-masked DPP move copies into a bank of the store register (CAP) and a second one replaces by the
// next 4 inputs (INJ):
//   before step k, k = 16q + 4b:  CAP  store_reg.bank[b] <- X.bank[3]        = y[16(q-1) + 4b .. +3]
//                                 INJ  X.bank[3]         <- xin(q).bank[b]   = x[16q + 4b .. +3]
//   lane 4b+i of xin / store_reg holds sample 16q + 4b + 3 - i, i.e. lane l <-> sample 16q + (l ^ 3)
// Post lane j (band j-1) filters sample k - j at step k; outside [0, n) it passes its input on.
//
// Pass-on lanes: the generic (first / last chunk) form selects, the asm form runs the arithmetic with
// (b0, b1) = (1, 0) and everything else 0: out = 1*x + 0 = x for every finite x.  Two consequences,
// both outside what audio hosts produce: a -0.0 sample comes out as +0.0, and a non-finite INPUT sample
// poisons the pass-on lanes, so up to 16 output samples BEFORE it are NaN as well (the reference's
// output is non-finite from that sample on, and so is ours).
constexpr int kRowRor1 = 0x121;
constexpr int kQuadIdentity = 0xE4;

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_mov_banks(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL,
                                                      0xf, BANK_MASK, false));
}

template <int B>
__device__ __forceinline__ void ring_port(float &store_reg, float &X, float xin)
{
    constexpr int cap_ctrl = (B == 3) ? kQuadIdentity : 0x120 + 4 * B + 4;
    constexpr int inj_ctrl = (B == 3) ? kQuadIdentity : 0x120 + 12 - 4 * B;
    store_reg = dpp_mov_banks<cap_ctrl, 1 << B>(store_reg, X);
    X = dpp_mov_banks<inj_ctrl, 0x8>(X, xin);
}

// per-lane constants of the ring form
struct RingLane {
    float pb0, pb1;         // pre:  (b0, b1) of band j, or (1, 0)
    float b2, a1, a2;       // post: band j-1, or 0
};

}  // namespace ohs
