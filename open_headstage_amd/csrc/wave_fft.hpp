// wave_fft.hpp -- one wavefront (64 lanes x 16 complex registers) computes one
// complex FFT-1024 entirely in registers + a private LDS slab of 8672 bytes.
//
// This is the transform behind rustfft's `Fft::process` call sites in the
// reference (src/dsp/convolution.rs:129,255,276): unnormalised, forward =
// e^{-2 pi i jk/N}.  It is a from-scratch gfx950 design, not a port:
//
//   1024 = 16 x 16 x 4.  A lane never exchanges data with a butterfly partner;
//   instead each lane runs whole radix-16 / radix-4 DFTs on its own registers and
//   the wave re-distributes points twice through LDS (T1, T2).  No workgroup
//   barrier is involved -- a wave is its own synchronisation domain.
//
//   time layout : lane m, reg a          holds z[64a + m]
//   bin  layout : lane L, reg 4r + j     holds Z[4L + j + 256 r]
//
//   forward: DFT16(a->p) . w1024^(m p) . T1 . DFT16(b->q) . w64^(c q) . T2 . DFT4(c->r)
//   inverse: the same graph run backwards with conjugated twiddles.
//
// LDS images (complex = 8 B units), conflict-free for ds_read_b64 (two 32-lane groups, 64 dword banks) and
// ds_write_b64 (four 16-lane groups, 32 dword banks) except a 2-way conflict on the inverse T2 write
// (tools/proto_wave_fft.py checks the index plan and the bank census on the CPU):
//   T1: addr = 68 p + m                        (m = 4b + c)
//   T2: addr = 65 (4 p_lo + c) + sigma(q, p_hi), sigma = 16(p_hi&1) + 8(p_hi>>1) + (q&7) + 32(q>>3)
// Padded, not XOR-swizzled: every access of an exchange is then ONE base register + an immediate offset.  (An
// unpadded 1024-entry XOR plan exists and is conflict-free everywhere, but needs a distinct address register
// per access; hipcc keeps them all live across the block loop and k_conv_p1 spilled 81 VGPRs.)
// The twiddle table w1024^(m p) is stored in the T1 layout as well, so the forward transform reads it by rows
// (lane m, reg p) and the inverse by columns (lane (p, c), reg b), both conflict-free -- one table, not two.
#pragma once
#include <hip/hip_runtime.h>

namespace ohs {

constexpr int kBlock = 512;
constexpr int kFft = 1024;
constexpr int kWaveLdsComplex = 15 * 68 + 64;   // per-wave LDS slab, in float2 units (8672 B): the last T1 row needs no padding,
                                                // the T2 image ends at 65 * 15 + 63

// twiddle tables kept in LDS per workgroup (float2 units), 8.5 KiB
constexpr int kTabA = 0;              // [16][68]  w1024^(m p) at t1_index(p, m): rows for the forward transform,
                                      //           columns (conjugated) for the inverse
// w64^(c q), 64 values, live in the four padding slots at the end of row q of that table (tabB_index): no bytes of
// their own -- which is what lets a 16th wave's slab fit into the CU's 160 KiB next to the P = 1 kernel's tables.
// forward: lane c = lane & 3, reg q (4 addresses per read: broadcast); inverse: lane q = lane >> 2, reg c (per
// 32 lanes 8 addresses in 8 different bank pairs)
constexpr int kTabComplex = 16 * 68;  // 1088 float2 = 8.5 KiB
__host__ __device__ __forceinline__ constexpr int tabB_index(int q, int c) { return 68 * q + 64 + c; }

// T1 image / twiddle table position of element (p, m), m = 4b + c
__host__ __device__ __forceinline__ constexpr int t1_index(int p, int m) { return 68 * p + m; }

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// Fused multiply-adds are written out (fmaf) and the files that include this header are built with
// -ffp-contract=off: which products get fused is then the same in every kernel that inlines these helpers,
// whatever the register pressure around them -- the single-stream paths (k_engine_call, k_engine_resident,
// k_conv_general) must agree bit for bit, and with implicit contraction they did so only by luck.
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
// a * conj(b)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -(a.x * b.y)));
}
// acc += a * b
__device__ __forceinline__ void cmac(float2 &acc, float2 a, float2 b)
{
    acc.x = fmaf(-a.y, b.y, fmaf(a.x, b.x, acc.x));
    acc.y = fmaf(a.y, b.x, fmaf(a.x, b.y, acc.y));
}
// acc += j * (a * b)
__device__ __forceinline__ void cmac_j(float2 &acc, float2 a, float2 b)
{
    acc.x = fmaf(-a.y, b.x, fmaf(-a.x, b.y, acc.x));
    acc.y = fmaf(-a.y, b.y, fmaf(a.x, b.x, acc.y));
}
template <bool CONJ>
__device__ __forceinline__ float2 ctw(float2 a, float2 w) { return CONJ ? cmulc(a, w) : cmul(a, w); }

// LDS write -> read hand-off inside one wave: DS operations of a wave execute in
// order, so only the compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS accessors of the FFT exchanges and table reads.  hipcc's load/store optimiser fuses two ds_read_b64
// of one base register into a ds_read2_b64, which the LDS serves at HALF the bytes per clock of two separate
// ds_read_b64 (MI355X: 2 x 32-lane groups per b64 read, 4 x 16 per half of a read2; measured by
// tools/ubench_lds.py).  A volatile access is left alone by that pass; program order among them is what the
// exchanges want anyway.
// OHS_TW_LD: a twiddle-table read; OHS_TW_PIN: the 15 reads of a stage stay one burst in front of their products
// (hipcc put each read in front of its multiplication and waited for it: eight exposed LDS round trips per stage)
#define OHS_TW_LD(p) lds_ld(p)
#define OHS_TW_PIN() __builtin_amdgcn_sched_barrier(0)
// (accessed as one 64-bit integer, not as a <2 x float> vector: with a vector type in play hipcc packs the
// following complex arithmetic into v_pk_* instructions and pays for it in register moves)
__device__ __forceinline__ float2 lds_ld(const float2 *p)
{
    const unsigned long long t = *(const volatile __attribute__((address_space(3))) unsigned long long *)p;
    return make_float2(__uint_as_float((unsigned)t), __uint_as_float((unsigned)(t >> 32)));
}
__device__ __forceinline__ void lds_st(float2 *p, float2 v)
{
    *(volatile __attribute__((address_space(3))) unsigned long long *)p =
        (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
}
// 16-byte LDS read (ds_read_b128: one instruction, one queue slot for two complex values)
typedef unsigned lds_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_ld2(const float4 *p, float2 &a, float2 &b)
{
    const lds_u32x4 t = *(const volatile __attribute__((address_space(3))) lds_u32x4 *)p;
    a = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
    b = make_float2(__uint_as_float(t.z), __uint_as_float(t.w));
}

// 4-point DFT in place: (a,b,c,d) = inputs 0..3 -> outputs 0..3
template <bool INV>
__device__ __forceinline__ void dft4(float2 &a, float2 &b, float2 &c, float2 &d)
{
    const float2 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = csub(b, d);
    // forward: -j*(b-d); inverse: +j*(b-d)
    const float2 jb = INV ? make_float2(-bmd.y, bmd.x) : make_float2(bmd.y, -bmd.x);
    a = cadd(apc, bpd);
    b = cadd(amc, jb);
    c = csub(apc, bpd);
    d = csub(amc, jb);
}

// multiply by w16^e (forward) or its conjugate (inverse), e a compile-time constant
template <bool INV, int E>
__device__ __forceinline__ float2 mul_w16(float2 a)
{
    constexpr float C1 = 0.92387953251128674f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508977f;  // sin(pi/8)
    constexpr float R2 = 0.70710678118654752f;  // sqrt(1/2)
    if constexpr (E == 0) return a;
    // forward twiddle w = (wr, -wi); inverse = (wr, +wi)
    else if constexpr (E == 1) return ctw<INV>(a, make_float2(C1, -S1));
    else if constexpr (E == 2)
        return INV ? make_float2((a.x - a.y) * R2, (a.x + a.y) * R2)
                   : make_float2((a.x + a.y) * R2, (a.y - a.x) * R2);
    else if constexpr (E == 3) return ctw<INV>(a, make_float2(S1, -C1));
    else if constexpr (E == 4) return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
    else if constexpr (E == 6)
        return INV ? make_float2((-a.x - a.y) * R2, (a.x - a.y) * R2)
                   : make_float2((a.y - a.x) * R2, (-a.x - a.y) * R2);
    else /* E == 9 */ return ctw<INV>(a, make_float2(-C1, S1));
}

// forward 4-point DFT whose inputs 2 and 3 are zero (the zero-padded upper half of a block, convolution.rs:248-254):
// 8 real additions instead of 16.  hipcc cannot do this by itself: x + 0.0 is not x under IEEE rules (-0.0).
__device__ __forceinline__ void dft4_fwd_upper_zero(float2 &a, float2 &b, float2 &c, float2 &d)
{
    const float2 a0 = a, b0 = b;
    a = cadd(a0, b0);
    b = make_float2(a0.x + b0.y, a0.y - b0.x);      // a - j b
    c = csub(a0, b0);
    d = make_float2(a0.x - b0.y, a0.y + b0.x);      // a + j b
}

// 16-point DFT on registers, natural order in, natural order out.  ZPAD: inputs 8..15 are zero.
template <bool INV, bool ZPAD = false>
__device__ __forceinline__ void dft16(float2 (&v)[16])
{
    // n = 4 n1 + n2 ; k = k1 + 4 k2
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        if constexpr (ZPAD && !INV) dft4_fwd_upper_zero(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
        else dft4<INV>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    }
    // now v[4 k1 + n2] = t[n2][k1]; twiddle by w16^(n2 k1)
    v[5] = mul_w16<INV, 1>(v[5]);
    v[6] = mul_w16<INV, 2>(v[6]);
    v[7] = mul_w16<INV, 3>(v[7]);
    v[9] = mul_w16<INV, 2>(v[9]);
    v[10] = mul_w16<INV, 4>(v[10]);
    v[11] = mul_w16<INV, 6>(v[11]);
    v[13] = mul_w16<INV, 3>(v[13]);
    v[14] = mul_w16<INV, 6>(v[14]);
    v[15] = mul_w16<INV, 9>(v[15]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4<INV>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    // v[4 k1 + k2] = X[k1 + 4 k2]  -> transpose the 4x4 register grid (pure renaming)
    float2 t;
    t = v[1];  v[1] = v[4];   v[4] = t;
    t = v[2];  v[2] = v[8];   v[8] = t;
    t = v[3];  v[3] = v[12];  v[12] = t;
    t = v[6];  v[6] = v[9];   v[9] = t;
    t = v[7];  v[7] = v[13];  v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

// The 8.5 KiB of per-workgroup twiddle tables are expanded ONCE per device on the host
// (expand_twiddle_tables, from exp(-2 pi i e / 1024) computed in f64 and rounded to f32, as rustfft
// does); a kernel only copies them into LDS with 16-byte loads.
__device__ __forceinline__ void fill_twiddle_tables(float2 *tab, const float2 *__restrict__ gtab,
                                                    int tid, int nthreads)
{
    const float4 *src = reinterpret_cast<const float4 *>(gtab);
    float4 *dst = reinterpret_cast<float4 *>(tab);
#pragma unroll 4
    for (int i = tid; i < kTabComplex / 2; i += nthreads) dst[i] = src[i];
}

// host side: expanded[kTabComplex] from tw[e] = exp(-2 pi i e / 1024)
inline void expand_twiddle_tables(const float2 *tw, float2 *expanded)
{
    for (int i = 0; i < kTabComplex; ++i) expanded[i] = tw[0];                                // padding
    for (int p = 0; p < 16; ++p)
        for (int m = 0; m < 64; ++m) expanded[kTabA + t1_index(p, m)] = tw[(m * p) & 1023];   // w1024^(m p)
    for (int q = 0; q < 16; ++q)
        for (int c = 0; c < 4; ++c) expanded[tabB_index(q, c)] = tw[(16 * c * q) & 1023];     // w64^(c q)
}

__host__ __device__ __forceinline__ constexpr int t2_sigma(int q, int p_hi)
{
    return 16 * (p_hi & 1) + 8 * (p_hi >> 1) + (q & 7) + 32 * (q >> 3);
}
// T2 image position of element (rr = 4 p_lo + c, q, p_hi)
__host__ __device__ __forceinline__ constexpr int t2_index(int rr, int sigma) { return 65 * rr + sigma; }

// time layout -> bin layout (unnormalised forward DFT).
// MIRROR: also return zm[reg] = Z[N - k(lane, reg)], the spectrum at the mirrored bin, which the packed
// two-channel formulation needs next to Z[k].  It costs no cross-lane traffic: the last stage is a DFT4
// over the T2 image, so a lane simply ALSO reads the four 4-point groups that produce its mirror bins --
// those of lane 63 - L (k = 4L + j + 256 r, j != 0  ->  N - k = 4(63 - L) + (4 - j) + 256 (3 - r)) and, for
// j == 0, of lane 64 - L -- and runs four more DFT4s: 16 ds_read_b64 + 64 adds instead of the 32
// ds_bpermute_b32 of wave_mirror (5.7 LDS cycles each, tools/ubench_lds.py).
// ZPAD: the caller guarantees v[8..15] == 0 on entry (a 512-frame block padded to 1024).
// forward transform up to and including the T2 write: on return the T2 image holds, at t2_index(4 j + c, sigma(L)),
// input c of the last stage's 4-point group (L, j) -- the group whose outputs are the bins 4 L + j + 256 r
template <bool ZPAD>
__device__ __forceinline__ void wave_fft_fwd_head(float2 (&v)[16], float2 *lds, const float2 *tab, int lane)
{
    dft16<false, ZPAD>(v);                                  // over a -> p
    {
        // all 15 table reads in one burst, pinned there: left alone, hipcc issues each read right in front of its
        // multiplication and waits for it -- eight exposed LDS round trips per twiddle stage
        float2 tw[16];
#pragma unroll
        for (int p = 1; p < 16; ++p) tw[p] = OHS_TW_LD(&tab[kTabA + t1_index(p, lane)]);
        OHS_TW_PIN();
#pragma unroll
        for (int p = 1; p < 16; ++p) v[p] = cmul(v[p], tw[p]);
    }
    const int pp = lane >> 2, c = lane & 3;
    const int t1r = t1_index(pp, c);                        // t1_index(pp, 4b + c) = t1r + 4b
#pragma unroll
    for (int p = 0; p < 16; ++p) lds_st(&lds[t1_index(p, lane)], v[p]);  // T1
    wave_lds_sync();
#pragma unroll
    for (int b = 0; b < 16; ++b) v[b] = lds_ld(&lds[t1r + 4 * b]);
    wave_lds_sync();
    dft16<false>(v);                                        // over b -> q
    {
        float2 tw[16];
#pragma unroll
        for (int q = 1; q < 16; ++q) tw[q] = OHS_TW_LD(&tab[tabB_index(q, lane & 3)]);
        OHS_TW_PIN();
#pragma unroll
        for (int q = 1; q < 16; ++q) v[q] = cmul(v[q], tw[q]);
    }
    const int p_lo = pp & 3, p_hi = pp >> 2;
    const int rrw = 4 * p_lo + c;
#pragma unroll
    for (int q = 0; q < 16; ++q) lds_st(&lds[t2_index(rrw, t2_sigma(q, p_hi))], v[q]);  // T2
    wave_lds_sync();
}

template <bool MIRROR, bool ZPAD>
__device__ __forceinline__ void wave_fft_fwd_impl(float2 (&v)[16], float2 (&zm)[16], float2 *lds, const float2 *tab,
                                                  int lane)
{
    wave_fft_fwd_head<ZPAD>(v, lds, tab, lane);
    const int rbase = t2_sigma(lane >> 2, lane & 3);
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) v[rr] = lds_ld(&lds[t2_index(rr, rbase)]);   // rr = 4 p_lo + c
    float2 m[16];
    if constexpr (MIRROR) {
        const int la = 63 - lane, lb = (64 - lane) & 63;
        const int rbA = t2_sigma(la >> 2, la & 3), rbB = t2_sigma(lb >> 2, lb & 3);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) m[cc] = lds_ld(&lds[t2_index(cc, rbB)]);               // group j' = 0 of lane 64 - L
#pragma unroll
        for (int rr = 4; rr < 16; ++rr) m[rr] = lds_ld(&lds[t2_index(rr, rbA)]);              // groups j' = 1..3 of lane 63 - L
    }
    wave_lds_sync();
    // DFT4 over c for each j = p_lo; result to reg 4 r + j
#pragma unroll
    for (int j = 0; j < 4; ++j) dft4<false>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    float2 t;   // (j, r) -> (r, j): 4x4 register transpose
    t = v[1];  v[1] = v[4];   v[4] = t;
    t = v[2];  v[2] = v[8];   v[8] = t;
    t = v[3];  v[3] = v[12];  v[12] = t;
    t = v[6];  v[6] = v[9];   v[9] = t;
    t = v[7];  v[7] = v[13];  v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
    if constexpr (MIRROR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dft4<false>(m[4 * j], m[4 * j + 1], m[4 * j + 2], m[4 * j + 3]);   // m[4 j' + r']
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 1; j < 4; ++j) zm[4 * r + j] = m[4 * (4 - j) + (3 - r)];
            // j == 0: lanes 1..63 take bin 256 (3 - r) of lane 64 - L; lane 0 (L = 0: N - 256 r) its own 256 ((4 - r) & 3)
            const float2 g = m[3 - r], own = m[(4 - r) & 3];
            zm[4 * r] = (lane == 0) ? own : g;
        }
    }
}

__device__ __forceinline__ void wave_fft_fwd(float2 (&v)[16], float2 *lds, const float2 *tab, int lane)
{
    float2 unused[16];
    wave_fft_fwd_impl<false, false>(v, unused, lds, tab, lane);
}
// the same for a zero-padded block: v[8..15] must be zero
__device__ __forceinline__ void wave_fft_fwd_zpad(float2 (&v)[16], float2 *lds, const float2 *tab, int lane)
{
    float2 unused[16];
    wave_fft_fwd_impl<false, true>(v, unused, lds, tab, lane);
}
// zero-padded block in, spectrum and mirror spectrum out (the P = 1 kernels)
__device__ __forceinline__ void wave_fft_fwd_mirror(float2 (&v)[16], float2 (&zm)[16], float2 *lds, const float2 *tab,
                                                    int lane)
{
    wave_fft_fwd_impl<true, true>(v, zm, lds, tab, lane);
}

__device__ __forceinline__ void wave_fft_inv_tail(float2 (&v)[16], float2 *lds, const float2 *tab, int lane);

// bin layout -> time layout (unnormalised inverse DFT)
__device__ __forceinline__ void wave_fft_inv(float2 (&v)[16], float2 *lds, const float2 *tab, int lane)
{
    float2 t;   // (r, j) -> (j, r)
    t = v[1];  v[1] = v[4];   v[4] = t;
    t = v[2];  v[2] = v[8];   v[8] = t;
    t = v[3];  v[3] = v[12];  v[12] = t;
    t = v[6];  v[6] = v[9];   v[9] = t;
    t = v[7];  v[7] = v[13];  v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
#pragma unroll
    for (int j = 0; j < 4; ++j) dft4<true>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);  // r -> c
#pragma unroll
    for (int c = 1; c < 4; ++c) {
        const float2 w = OHS_TW_LD(&tab[tabB_index(lane >> 2, c)]);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * j + c] = cmulc(v[4 * j + c], w);
    }
    const int rbase = t2_sigma(lane >> 2, lane & 3);
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) lds_st(&lds[t2_index(rr, rbase)], v[rr]);     // T2'
    wave_lds_sync();
    wave_fft_inv_tail(v, lds, tab, lane);
}

// inverse transform from the T2' image on (the counterpart of wave_fft_fwd_head)
__device__ __forceinline__ void wave_fft_inv_tail(float2 (&v)[16], float2 *lds, const float2 *tab, int lane)
{
    const int pp = lane >> 2, c = lane & 3;
    const int p_lo = pp & 3, p_hi = pp >> 2;
    const int rrw = 4 * p_lo + c;
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = lds_ld(&lds[t2_index(rrw, t2_sigma(q, p_hi))]);
    wave_lds_sync();
    dft16<true>(v);                                         // over q -> b
    const int t1r = t1_index(pp, c);                        // t1_index(pp, 4b + c) = t1r + 4b
    {
        float2 tw[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) tw[b] = OHS_TW_LD(&tab[kTabA + t1r + 4 * b]);
        OHS_TW_PIN();
#pragma unroll
        for (int b = 0; b < 16; ++b) v[b] = cmulc(v[b], tw[b]);   // w1024^((4b+c) p), conj
    }
#pragma unroll
    for (int b = 0; b < 16; ++b) lds_st(&lds[t1r + 4 * b], v[b]);    // T1'
    wave_lds_sync();
#pragma unroll
    for (int p = 0; p < 16; ++p) v[p] = lds_ld(&lds[t1_index(p, lane)]);
    wave_lds_sync();
    dft16<true>(v);                                         // over p -> a
}

// ---------------------------------------------------------------------------------
// Paired bin layout (k_conv_p1): bins k and N - k in the SAME lane, so that W = Z C + conj(Z[N-k]) D needs no
// mirror spectrum -- none of the 16 extra ds_read_b64 and four extra DFT4s of wave_fft_fwd_mirror.  The last
// forward stage is a DFT4 per group (L', j) -> bins 4 L' + j + 256 r, and the mirror of group (L', j) is the group
// (63 - L', 4 - j) with r reversed (j = 0: (64 - L', 0); (0, 0) and (32, 0) mirror themselves).  Lane L therefore takes
// four groups, slot g -> register 4 g + r:
//     g = 0: (L, 1)          g = 1: (63 - L, 3)                               mirrors of each other, r <-> 3 - r
//     g = 2: (L, 2)          g = 3: (63 - L, 2)          lanes  0..31         mirrors of each other, r <-> 3 - r
//     g = 2: (L, 0)          g = 3: (64 - L, 0)          lanes 33..63         mirrors of each other, r <-> 3 - r
//     g = 2: (32, 0)         g = 3: (0, 0)               lane 32              each its own mirror (r <-> 3 - r, r <-> (4 - r) & 3)
// Per instruction and 32-lane half the group index j is uniform and the lanes L' are a permutation of one half (per
// 16 lanes of a write: of a set with the same bank census), so the bank behaviour is that of the natural layout
// (tools/proto_wave_fft.py checks coverage, mirror map, numerics and the census).  The inverse transform pays with
// w64 rows per slot -- slots 0 and 2 hold groups of the lane itself (row L >> 2), slots 1 and 3 have rows of their
// own: 9 table reads instead of 3.  The arithmetic per bin is that of the natural layout, operation for operation:
// results are bit-identical.
// ---------------------------------------------------------------------------------
__host__ __device__ __forceinline__ constexpr void paired_group(int lane, int g, int &lg, int &j)
{
    if (g == 0) { lg = lane; j = 1; }
    else if (g == 1) { lg = 63 - lane; j = 3; }
    else if (lane < 32) { lg = g == 2 ? lane : 63 - lane; j = 2; }
    else { lg = g == 2 ? lane : (64 - lane) & 31; j = 0; }      // (lane 32: slot 3 = group (0, 0))
}
// position in the natural bin layout (register * 64 + lane) of the bin that paired register 4 g + r of `lane` holds
__host__ __device__ __forceinline__ constexpr int paired_to_natural(int lane, int reg)
{
    int lg = 0, j = 0;
    paired_group(lane, reg >> 2, lg, j);
    return (4 * (reg & 3) + j) * 64 + lg;
}
// per-lane constants of the paired layout: T2-image offset of a slot's element c = 0 (element c: + 65 c) and the
// slot's row of the w64 table (entry c: + c)
struct PairedPlan {
    int t2[4];
    int tw[4];
};
__device__ __forceinline__ PairedPlan paired_plan(int lane)
{
    PairedPlan p;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int lg = 0, j = 0;
        paired_group(lane, g, lg, j);
        p.t2[g] = t2_index(4 * j, t2_sigma(lg >> 2, lg & 3));
        p.tw[g] = tabB_index(lg >> 2, 0);
    }
    return p;
}

// zero-padded block in (v[8..15] == 0), spectrum in the paired layout out
__device__ __forceinline__ void wave_fft_fwd_paired(float2 (&v)[16], float2 *lds, const float2 *tab, int lane,
                                                    const PairedPlan &pl)
{
    wave_fft_fwd_head<true>(v, lds, tab, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * g + c] = lds_ld(&lds[pl.t2[g] + 65 * c]);
    }
    wave_lds_sync();
#pragma unroll
    for (int g = 0; g < 4; ++g) dft4<false>(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);     // c -> r
}

// the paired register of the same lane that holds bin N - k for the bin k of register 4 g + r
__host__ __device__ __forceinline__ constexpr int paired_mirror_reg(int g, int r, bool lane32)
{
    if (g == 0) return 4 + (3 - r);
    if (g == 1) return 3 - r;
    if (lane32) return 4 * g + (g == 2 ? 3 - r : (4 - r) & 3);
    return (g == 2 ? 12 : 8) + (3 - r);
}
// compile-time check of the plan: every DFT4 group in exactly one slot, every mirror in the lane that needs it
constexpr bool paired_layout_ok()
{
    bool seen[256] = {};
    for (int lane = 0; lane < 64; ++lane)
        for (int g = 0; g < 4; ++g) {
            int lg = 0, j = 0;
            paired_group(lane, g, lg, j);
            if (lg < 0 || lg > 63 || seen[4 * lg + j]) return false;
            seen[4 * lg + j] = true;
            for (int r = 0; r < 4; ++r) {
                const int m = paired_mirror_reg(g, r, lane == 32);
                int lm = 0, jm = 0;
                paired_group(lane, m >> 2, lm, jm);
                if ((4 * lg + j + 256 * r + 4 * lm + jm + 256 * (m & 3)) % kFft != 0) return false;
            }
        }
    return true;
}
static_assert(paired_layout_ok(), "paired bin layout: coverage / mirror map");

// the mirror value Z[N - k] of paired register 4 g + r, from the lane's own registers
__device__ __forceinline__ float2 paired_mirror(const float2 (&z)[16], int g, int r, bool lane32)
{
    if (g < 2) return z[paired_mirror_reg(g, r, false)];
    const float2 other = z[paired_mirror_reg(g, r, false)], self = z[paired_mirror_reg(g, r, true)];
    return make_float2(lane32 ? self.x : other.x, lane32 ? self.y : other.y);
}

// paired layout in, time layout out (unnormalised inverse DFT)
__device__ __forceinline__ void wave_fft_inv_paired(float2 (&v)[16], float2 *lds, const float2 *tab, int lane,
                                                    const PairedPlan &pl)
{
#pragma unroll
    for (int g = 0; g < 4; ++g) dft4<true>(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);      // r -> c
    {
        float2 tw[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g == 2) continue;       // the lane's own row again: slot 0's values
#pragma unroll
            for (int c = 1; c < 4; ++c) tw[4 * g + c] = OHS_TW_LD(&tab[pl.tw[g] + c]);
        }
        OHS_TW_PIN();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int c = 1; c < 4; ++c) v[4 * g + c] = cmulc(v[4 * g + c], tw[4 * (g == 2 ? 0 : g) + c]);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int c = 0; c < 4; ++c) lds_st(&lds[pl.t2[g] + 65 * c], v[4 * g + c]);                    // T2'
    }
    wave_lds_sync();
    wave_fft_inv_tail(v, lds, tab, lane);
}

// Fetch, for every register of the bin layout, the value at bin (N - k).
__device__ __forceinline__ void wave_mirror(const float2 (&z)[16], float2 (&zm)[16], int lane)
{
    const int srcA = (63 - lane) << 2;          // j = 1..3 : lane 63-L
    const int srcB = ((64 - lane) & 63) << 2;   // j = 0    : lane 64-L (lane 0 -> itself)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            const float2 s = z[4 * (3 - r) + (4 - j)];
            zm[4 * r + j].x = __int_as_float(__builtin_amdgcn_ds_bpermute(srcA, __float_as_int(s.x)));
            zm[4 * r + j].y = __int_as_float(__builtin_amdgcn_ds_bpermute(srcA, __float_as_int(s.y)));
        }
        // j == 0: lanes 1..63 take reg 4(3-r) of lane 64-L; lane 0 takes its own reg 4((4-r)&3)
        const float2 s = z[4 * (3 - r)];
        const float2 own = z[4 * ((4 - r) & 3)];
        float2 g;
        g.x = __int_as_float(__builtin_amdgcn_ds_bpermute(srcB, __float_as_int(s.x)));
        g.y = __int_as_float(__builtin_amdgcn_ds_bpermute(srcB, __float_as_int(s.y)));
        zm[4 * r] = (lane == 0) ? own : g;
    }
}

}  // namespace ohs
