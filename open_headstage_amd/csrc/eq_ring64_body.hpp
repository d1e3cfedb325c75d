// eq_ring64_body.hpp -- the ring-form DF2T cascade (<= 12 bands) with the ring closed over ALL 64 lanes of a wave: ONE
// chain per wave.  Device code of k_eq_ring's second form (eq_kernels.hip), chosen where the chip has a SIMD to spare
// for every chain (BASELINE configs[2]: 512 chains, 1 024 SIMDs).
// Every file that includes this header MUST be built with -ffp-contract=off (parametric_eq.rs:116-122: every product and
// sum of the recurrence rounds by itself).
//
// Why.  The ring form's time is its VALU instruction count: a lone wave issues one instruction per ~4.1 cycles whatever
// its lanes do.  In a 16-lane DPP row (eq_ring2_body.hpp) a chain has 10 bands + 6 pass-on lanes, and the
// I/O port -- capture 4 outputs, inject 4 inputs -- runs every 4 steps: 6 + 2 / 4 = 6.5 instructions per sample.  gfx950
// still executes gfx9's wave-wide DPP controls at the row controls' cost (wave_ror:1; tools/ubench_wave_dpp.hip,
// profiles/r05_ubench_wave_dpp.txt).  With the ring closed over the wave, lanes 13 .. 63 pass samples on, rows 1 .. 3
// are a conveyor of 48 samples, and the port -- two moves with row_mask 0xe, one store, one load -- runs every 48 steps:
// 6 + 2 / 48 = 6.04 instructions per sample (prototype: tools/proto_eq_wave_ring.py, 11.02 against 11.49 ns).
//
// Lane roles and arithmetic are eq_ring2_body.hpp's (band L: pre lane L, post lane L + 1; O T A P N M per step), with
// wave_ror:1 where that form has row_ror:1.  What differs is how a launch starts and ends.  The same six instructions
// run on every lane from the first step to the last -- no step is gated.  Instead
//   head   the ring starts empty (X = 0, state = 0: a band that sees zeros in a zero state stays at zero), and band L's
//          state from the previous launch is put into its post lane behind step L + 1, the step in front of the one in
//          which its first sample arrives;
//   tail   band L's state is taken out of its post lane behind step n + L + 1, the step in which it filters sample
//          n - 1; the ring then runs on zeros until the last outputs have travelled into the conveyor and been stored.
// Timeline (steps count from 1; lane 63 - i of rows 1 .. 3 holds sample 48 g + i when group g is injected, behind step 48 g):
//   sample i is in lane 0 at step i + 1, filtered by band L (post lane L + 1) at step i + L + 2, whatever the number of
//   bands it reaches lane 16 + m at step i + 17 + m; the port behind step 48 (g + 1) finds sample 48 g + 47 - l in lane
//   l >= 16: outputs leave 16 samples behind the inputs that replace them.
#pragma once
#include "kernels.h"
#include "eq_ring2_body.hpp"    // v2f, dpp helpers, RingLane, ring2_ld / ring2_st

namespace ohs {

constexpr int kWaveRor1 = 0x13C;        // DPP_WF_RR1: lane l <- lane l - 1, lane 0 <- lane 63
constexpr int kR64Group = 48;           // samples per group = lanes of the conveyor

struct Ring64Regs {
    float X, st;        // v2, v3
    v2f u;              // v[4:5]   (b0, b1) * X of the pre lane
    float b2x;          // v7       b2 * X(wave_ror:1) of the post lane
    v2f s;              // v[10:11] (s1, s2)
};

// rows 1 .. 3 of `src` into `old` (row 0 keeps `old`)
__device__ __forceinline__ float ring64_rows123(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), kQuadIdentity, 0xe, 0xf, false));
}

// P and M: what a step leaves behind for the next one
__device__ __forceinline__ void ring64_pm(Ring64Regs &r, const RingLane &c)
{
    r.u = (v2f){c.pb0 * r.X, c.pb1 * r.X};
    r.b2x = c.b2 * dpp_mov<kWaveRor1>(r.X, r.X);
}

#define R64_FULL "row_mask:0xf bank_mask:0xf\n"
#define R64_O "v_add_f32_dpp v2, v4, v10 wave_ror:1 " R64_FULL
#define R64_T "v_add_f32_dpp v6, v5, v11 wave_ror:1 " R64_FULL
#define R64_A "v_pk_mul_f32 v[8:9], v[14:15], v[2:3] op_sel_hi:[1,0]\n"
#define R64_P "v_pk_mul_f32 v[4:5], v[12:13], v[2:3] op_sel_hi:[1,0]\n"
#define R64_N "v_pk_add_f32 v[10:11], v[6:7], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n"
#define R64_M "v_mul_f32_dpp v7, v2, v1 wave_ror:1 " R64_FULL
#define R64_STEP R64_O R64_T R64_A R64_P R64_N R64_M
#define R64_X2(B) B B
#define R64_X4(B) R64_X2(B) R64_X2(B)
#define R64_X8(B) R64_X4(B) R64_X4(B)
#define R64_X16(B) R64_X8(B) R64_X8(B)
#define R64_X32(B) R64_X16(B) R64_X16(B)
#define R64_STEPS47 R64_X32(R64_STEP) R64_X8(R64_STEP) R64_X4(R64_STEP) R64_X2(R64_STEP) R64_STEP
#define R64_ROWS123 "quad_perm:[0,1,2,3] row_mask:0xe bank_mask:0xf\n"

// One interior group as asm text: 47 plain steps, the port step (the input register REG is injected and refilled in place).
#define R64_GROUP(REG) \
        R64_STEPS47 \
        R64_O R64_T R64_A \
        "s_waitcnt vmcnt(2)\n" \
        "s_nop 0\n" \
        "v_mov_b32_dpp v3, v2 " R64_ROWS123 \
        "v_mov_b32_dpp v2, " REG " " R64_ROWS123 \
        "s_mov_b32 exec_lo, %[mlo]\n" \
        "v_min_u32_e32 v18, %[maxo], v0\n" \
        "global_store_dword v0, v3, %[dst] offset:-64\n" \
        "global_load_dword " REG ", v18, %[src] offset:576\n" \
        "s_mov_b32 exec_lo, -1\n" \
        "v_add_u32_e32 v0, %[kg], v0\n" \
        R64_P R64_N R64_M

// `pairs` >= 1 PAIRS of interior groups in ONE statement: every sample they filter and store exists, no band starts or
// ends inside them.  The input register is double-buffered (v16: even groups, v17: odd ones): the port of group g injects
// group g + 1, requested at port g - 2, and requests group g + 3 into the register it has just emptied -- 96 steps =
// ~1.06 us between a load and its use (with one register and 48 steps the wave waited for memory at every port: the
// prototype, alone on the chip, 11.05 -> 10.88 ns per sample; profiles/r05_proto_eq_wave_ring_ahead.txt).  In front of a
// port the store and the load of the previous port may still be in flight: vmcnt(2) (they retire in issue order).
// Loads may reach beyond the launch's last sample: the address is clamped to that sample's (`max_load_off`), so what
// arrives there is a copy of it instead of the zero the C++ form puts -- harmless: every band's state has left its lane
// before such a value reaches it, and outputs beyond n are never stored.
// On entry: the first step's P and M have run (by the C++ form), v16 / v17 = the inputs of the two groups BEHIND the first
// one here, v0 = this lane's byte offset of the first group's input slot; on exit v16 = the inputs of the group behind the
// next one.  Every 8-byte encoding is 8-byte aligned (a lone wave pays for one that straddles a fetch line); the
// vector-memory instructions run with row 0 masked.
__device__ __forceinline__ void eq_ring64_groups(const RingLane &c, Ring64Regs &r, float &xnext, float xnext2, unsigned &lane_off,
                                                 const float *src, float *dst, int pairs, unsigned max_load_off)
{
    const v2f pb01 = {c.pb0, c.pb1}, a12 = {c.a1, c.a2};
    unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane(pairs);
    unsigned kg = kR64Group * 4;                        // (not const: the host pass of hipcc refuses constants as asm operands)
    unsigned mlo = 0xFFFF0000u;                         // EXEC's low half with row 0 off (a 64-bit literal would reach the
                                                        // scalar unit as its low 32 bits, zero-extended: rows 2 and 3 off as well)
    asm volatile(
        ".p2align 5\n"
        "s_nop 1\n"
        "s_nop 0\n"
        "1:\n"
        R64_GROUP("v16")
        R64_GROUP("v17")
        "s_sub_u32 %[cnt], %[cnt], 1\n"
        "s_cmp_lg_u32 %[cnt], 0\n"
        "s_cbranch_scc1 1b\n"
        "s_nop 0\n"
        "s_waitcnt vmcnt(0)\n"
        "s_nop 1\n"
        : [X] "+{v2}"(r.X), [st] "+{v3}"(r.st), [u] "+{v[4:5]}"(r.u), [b2x] "+{v7}"(r.b2x), [s] "+{v[10:11]}"(r.s),
          [voff] "+{v0}"(lane_off), [xa] "+{v16}"(xnext), [xb] "+{v17}"(xnext2), [cnt] "+s"(cnt)
        : [b2] "{v1}"(c.b2), [pb01] "{v[12:13]}"(pb01), [a12] "{v[14:15]}"(a12), [src] "s"(src), [dst] "s"(dst), [kg] "s"(kg),
          [mlo] "s"(mlo), [maxo] "s"(max_load_off)
        : "v6", "v8", "v9", "v18", "memory", "scc");
}

// PER_STREAM: the chain's stream owns its bands (parametric_eq.rs:125-129) -- constants, state slots and the NUMBER of enabled
// bands come from stabs[chain / 2] (kernels.h: EqStreamTable) instead of the launch's one table.
template <bool PER_STREAM>
__device__ __forceinline__ void eq_ring64_wave_t(const float *in, float *out, long long stream_stride, long long ch_stride,
                                                 long long n, int n_chains, int nb_shared, const EqPassTable &tab,
                                                 const EqStreamTable *__restrict__ stabs, float *__restrict__ state, long long chain)
{
    constexpr int G = kR64Group;
    if (chain >= n_chains) return;
    const int lane = threadIdx.x & 63;
    const int n32 = (int)n;
    int nb = nb_shared;
    if constexpr (PER_STREAM) nb = stabs[chain >> 1].nb;
    const bool pre = lane < nb, band = lane >= 1 && lane <= nb;
    const int jb = band ? lane - 1 : 0, jp = pre ? lane : 0;
    RingLane c;
    int slot;
    if constexpr (PER_STREAM) {
        const EqStreamTable *T = stabs + (chain >> 1);
        c.pb0 = pre ? T->b0[jp] : 1.0f;
        c.pb1 = pre ? T->b1[jp] : 0.0f;
        c.b2 = band ? T->b2[jb] : 0.0f;
        c.a1 = band ? T->a1[jb] : 0.0f;
        c.a2 = band ? T->a2[jb] : 0.0f;
        slot = T->slot[jb];
    } else {
        c.pb0 = pre ? tab.b0[jp] : 1.0f;
        c.pb1 = pre ? tab.b1[jp] : 0.0f;
        c.b2 = band ? tab.b2[jb] : 0.0f;
        c.a1 = band ? tab.a1[jb] : 0.0f;
        c.a2 = band ? tab.a2[jb] : 0.0f;
        slot = tab.slot[jb];
    }
    // wave-uniform bases + per-lane 32-bit byte offsets (rows 1 .. 3: lane 63 - i <-> sample i of a group)
    const long long base = (chain >> 1) * stream_stride + (chain & 1) * ch_stride;
    const float *src0 = in + base;
    float *dst0 = out + base;
    float *state0 = state + chain * (kEqStateSlots * 2);
    const unsigned state_off = (unsigned)(slot * 2) * 4u;
    const bool conv = lane >= 16;
    unsigned lane_off = (unsigned)(63 - lane) * 4u;
    v2f s_init = {0.0f, 0.0f};
    if (band) { s_init.x = ring2_ld(state0, state_off); s_init.y = ring2_ld(state0, state_off + 4u); }
    v2f s_save = s_init;
    auto load_group = [&](int g) -> float {         // group g's inputs, zeros beyond n
        const int i = g * G + 63 - lane;
        return (conv && i < n32) ? ring2_ld(src0, (unsigned)i * 4u) : 0.0f;
    };
    Ring64Regs r;
    r.st = 0.0f; r.b2x = 0.0f; r.u = (v2f){0.0f, 0.0f}; r.s = (v2f){0.0f, 0.0f};
    const float x0 = load_group(0);
    float xnext = load_group(1);
    r.X = conv ? x0 : 0.0f;
    ring64_pm(r, c);
    // one group in the C++ form: the launch's first group (the bands' states arrive), its last ones (they leave; loads
    // and stores are checked against n)
    auto group_cpp = [&](int g) {
        auto step = [&](int k, bool port) {
            const int stp = g * G + k + 1;
            const float o = dpp_mov<kWaveRor1>(r.u.x, r.u.x) + r.s.x;           // O
            const float t2 = dpp_mov<kWaveRor1>(r.u.y, r.u.y) + r.s.y;          // T
            r.X = o;
            const v2f ao = {c.a1 * o, c.a2 * o};                                // A
            if (port) {
                r.st = ring64_rows123(r.st, r.X);
                r.X = ring64_rows123(r.X, xnext);
                const int yi = g * G + 47 - lane;
                if (conv && (unsigned)yi < (unsigned)n32) ring2_st(dst0, (unsigned)yi * 4u, r.st);
                xnext = load_group(g + 2);
            }
            const v2f sn = {t2 - ao.x, r.b2x - ao.y};                           // N (with the previous step's M)
            ring64_pm(r, c);                                                    // P, M
            r.s = sn;
            if (band && lane == stp) r.s = s_init;          // behind step L + 1: band L's first sample is next
            if (band && lane == stp - n32) s_save = r.s;    // behind step n + L + 1: band L has filtered sample n - 1
        };
#pragma unroll 1
        for (int k = 0; k < G - 1; ++k) step(k, false);
        step(G - 1, true);
    };
    const int g_total = (n32 + 16 + G - 1) / G;
    group_cpp(0);
    int g = 1;
    // groups 1 .. n / 48 - 1, two at a time (an odd one left over runs in the C++ form): every step filters existing samples
    // (the last of them ends at step 48 (n / 48) <= n: the first state leaves behind step n + 1), every store lands below n
    const int n_pairs = n32 >= 4 * G ? (n32 / G - 1) / 2 : 0;
    if (n_pairs >= 1) {
        const float xnext2 = load_group(3);         // (exists: n >= 192)
        lane_off += (unsigned)(G * 4);
        eq_ring64_groups(c, r, xnext, xnext2, lane_off, src0, dst0, n_pairs, (unsigned)(n32 - 1) * 4u - 576u);
        g += 2 * n_pairs;
    }
    for (; g < g_total; ++g) group_cpp(g);
    if (band) {
        ring2_st(state0, state_off, s_save.x);
        ring2_st(state0, state_off + 4u, s_save.y);
    }
}

// the shared-table form
__device__ __forceinline__ void eq_ring64_wave(const float *in, float *out, long long stream_stride, long long ch_stride,
                                               long long n, int n_chains, int nb, const EqPassTable &tab,
                                               float *__restrict__ state, long long chain)
{
    eq_ring64_wave_t<false>(in, out, stream_stride, ch_stride, n, n_chains, nb, tab, nullptr, state, chain);
}

}  // namespace ohs
