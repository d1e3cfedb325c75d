#!/usr/bin/env python3
"""Prototype (not product): the EQ's ring form over ALL 64 lanes of a wave -- one chain per wave.

k_eq_ring keeps a chain in a 16-lane DPP row (row_ror:1): 10 bands + 6 pass-on lanes, four chains per wave, and pays
6 VALU per step + an I/O port (2 moves) every 4 steps = 6.5 VALU per sample, which IS its time: a lone wave issues one
VALU instruction per ~4.1 cycles whatever its lanes do.  gfx950 still executes gfx9's wave-wide DPP controls
(wave_ror:1 at row_ror:1's cost: tools/ubench_wave_dpp.hip).  With the ring closed over the whole wave a chain has 53
pass-on lanes: rows 1..3 are a conveyor of 48 samples, and the port -- capture 48 outputs, inject 48 inputs: two moves
with row_mask 0xe -- runs every 48 steps: 6.04 VALU per sample.  At BASELINE configs[2] (512 chains, 1 024 SIMDs) a
chain per wave fits the chip, so the headline's step would shrink by what the port costs today.

This script writes the steady-state loop as a stand-alone kernel (same six instructions per step, same order, same
pinned registers as tools/gen_eq_ring2_asm.py; wave_ror:1 for row_ror:1), builds tools/bin/proto_eq_wave_ring and that
program (on the GPU box)
  * checks the outputs of every chain BIT FOR BIT against the host's DF2T cascade (parametric_eq.rs:116-122 order, every
    product and sum rounded separately: built with -ffp-contract=off), and
  * times 512 chains x 480 256 samples (one wave per workgroup) beside a lone wave's cycles per sample.
Head and tail of a launch, state hand-over between launches, per-stream tables and fewer than 48 samples are the
product's business and not modelled: the stream starts from zero state and is a multiple of 48 samples long.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
G = 48                      # samples per group = conveyor length (rows 1..3)
FULL = "row_mask:0xf bank_mask:0xf"
ROWS123 = "quad_perm:[0,1,2,3] row_mask:0xe bank_mask:0xf"

# PROTO_VARIANT: "wave" (default) all three DPP instructions wave_ror:1; "o" only O -- T and M feed the band lanes, which all
# sit in row 0 and read their left neighbour in row 0 (a pass-on lane's T / M results are zeros whichever lane it reads);
# "row" all three row_ror:1 (NOT a conveyor: wrong results, the timing reference of this loop)
VARIANT = os.environ.get("PROTO_VARIANT", "wave")
RO = "row_ror:1" if VARIANT == "row" else "wave_ror:1"
RT = "wave_ror:1" if VARIANT == "wave" else "row_ror:1"
O = f"v_add_f32_dpp v2, v4, v10 {RO} {FULL}"
T = f"v_add_f32_dpp v6, v5, v11 {RT} {FULL}"
A = "v_pk_mul_f32 v[8:9], v[14:15], v[2:3] op_sel_hi:[1,0]"
P = "v_pk_mul_f32 v[4:5], v[12:13], v[2:3] op_sel_hi:[1,0]"
N = "v_pk_add_f32 v[10:11], v[6:7], v[8:9] neg_lo:[0,1] neg_hi:[0,1]"
M = f"v_mul_f32_dpp v7, v2, v1 {RT} {FULL}"


# PROTO_AHEAD=2: the input register is double-buffered (v16 / v17, two groups per loop iteration): a load is consumed two
# ports = 96 steps = ~1.06 us after its issue instead of 48 steps = ~0.53 us
AHEAD = int(os.environ.get("PROTO_AHEAD", "1"))


def group(reg="v16", wait="s_waitcnt vmcnt(0)", load_off=2 * G * 4):
    out = []
    for k in range(G):
        out += [O, T, A]
        if k == G - 1:
            # the port, in front of the NEXT step's P and M: the 48 outputs that have travelled into rows 1..3 leave, the
            # next 48 inputs take their place; the store and the reload of the input register go out back to back
            out += [wait,                                       # the store and the load issued one group ago (they retire in issue order)
                    "s_nop 0",
                    f"v_mov_b32_dpp v3, v2 {ROWS123}",
                    f"v_mov_b32_dpp v2, {reg} {ROWS123}",
                    "global_store_dword v0, v3, %[dst]",
                    f"global_load_dword {reg}, v0, %[src] offset:{load_off}",
                    "v_add_u32_e32 v0, %[kg], v0",
                    "s_nop 0"]
        out += [P, N, M]
    return out


def main():
    if AHEAD == 2:
        # port g injects group g + 1 (loaded at port g - 2) and requests group g + 3 into the register it has just emptied;
        # in front of it the store and the load of port g - 1 may still be in flight: vmcnt(2)
        two = group("v16", "s_waitcnt vmcnt(2)", 3 * G * 4) + group("v17", "s_waitcnt vmcnt(2)", 3 * G * 4)
        body = [".p2align 5", "1:"] + two + ["s_sub_u32 %[cnt], %[cnt], 2", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 1b", "s_waitcnt vmcnt(0)", "s_nop 1"]
    else:
        body = [".p2align 5", "1:"] + group() + ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 1b", "s_waitcnt vmcnt(0)", "s_nop 1"]
    asm = "\n".join('        "' + l + '\\n"' for l in body)
    n_valu = sum(1 for l in group() if l.startswith("v_") and not l.startswith("v_add_u32"))
    src = r'''
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int G = %(G)d, NB = 10;
struct Coef { float b0[NB], b1[NB], b2[NB], a1[NB], a2[NB]; };

// one chain per wave (= per workgroup); chain c: src + c * stride, n = groups * G samples, zero state at entry
__global__ __launch_bounds__(64) void k(const float *src, float *dst, long long stride, int groups, Coef cf, unsigned long long *ticks)
{
    const int lane = threadIdx.x;
    // band L: pre lane L, post lane L + 1;  lanes 11 .. 63 pass on
    const bool pre = lane < NB, post = lane >= 1 && lane <= NB;
    v2f pb01 = {pre ? cf.b0[lane] : 1.0f, pre ? cf.b1[lane] : 0.0f};
    v2f a12 = {post ? cf.a1[lane - 1] : 0.0f, post ? cf.a2[lane - 1] : 0.0f};
    float b2 = post ? cf.b2[lane - 1] : 0.0f;
    // rows 1 .. 3 are the conveyor: lane 63 holds the group's first sample, lane 16 its last; row 0's lanes point at a
    // scratch copy of the stream behind it (their loads and stores are never used)
    const long long base = (long long)blockIdx.x * stride;
    unsigned off = lane >= 16 ? (unsigned)((base + (63 - lane)) * 4) : (unsigned)((base + stride / 2 + lane) * 4);
    float X = 0.f, st = 0.f, b2x = 0.f, xin;
    v2f u = {0.f, 0.f}, s = {0.f, 0.f};
    unsigned cnt = (unsigned)groups, kg = G * 4;
    // prologue: group 0's inputs into the conveyor, group 1's requested; the first step's P and M
    xin = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(src) + off);
    float xin1 = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(src) + off + G * 4);
    float xin2 = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(src) + off + 2 * G * 4);
    X = lane >= 16 ? xin : 0.0f;
    // dst is written one group late (the outputs captured at the end of group g are those that entered rows 1 .. 3
    // during it); the host finds the lag
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(
        ".p2align 5\n"
        "s_nop 4\n"
        "s_nop 0\n"
        "%(P)s\n"
        "%(M)s\n"
        "s_nop 1\n"
        "s_nop 0\n"
%(asm)s
        : [X] "+{v2}"(X), [st] "+{v3}"(st), [u] "+{v[4:5]}"(u), [b2x] "+{v7}"(b2x), [s] "+{v[10:11]}"(s), [voff] "+{v0}"(off),
          [xin] "+{v16}"(xin1), [xin2] "+{v17}"(xin2), [cnt] "+s"(cnt)
        : [b2] "{v1}"(b2), [pb01] "{v[12:13]}"(pb01), [a12] "{v[14:15]}"(a12), [src] "s"(src), [dst] "s"(dst), [kg] "s"(kg)
        : "v6", "v8", "v9", "memory", "scc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && ticks) ticks[blockIdx.x] = t1 - t0;
}

static void host_cascade(const Coef &c, const std::vector<float> &x, std::vector<float> &y)
{
    float s1[NB] = {0}, s2[NB] = {0};
    y.resize(x.size());
    for (size_t i = 0; i < x.size(); ++i) {
        float v = x[i];
        for (int b = 0; b < NB; ++b) {      // DF2T, parametric_eq.rs:116-122 / biquad 0.4.2: each operation rounds by itself
            const float out = s1[b] + c.b0[b] * v;
            s1[b] = (s2[b] + c.b1[b] * v) - c.a1[b] * out;
            s2[b] = c.b2[b] * v - c.a2[b] * out;
            v = out;
        }
        y[i] = v;
    }
}

int main()
{
    Coef c;
    for (int b = 0; b < NB; ++b) {          // peaking sections, RBJ, 48 kHz (any stable table does)
        const double f = 63.0 * std::pow(2.0, b * 0.85), q = 1.41, g = (b & 1) ? 2.5 : -3.0;
        const double A_ = std::pow(10.0, g / 40.0), w = 2.0 * M_PI * f / 48000.0, al = std::sin(w) / (2.0 * q), a0 = 1.0 + al / A_;
        c.b0[b] = (float)((1.0 + al * A_) / a0); c.b1[b] = (float)(-2.0 * std::cos(w) / a0); c.b2[b] = (float)((1.0 - al * A_) / a0);
        c.a1[b] = (float)(-2.0 * std::cos(w) / a0); c.a2[b] = (float)((1.0 - al / A_) / a0);
    }
    const int chains = 512, groups = (480256 / G + 2) & ~1;        // an even number of groups (PROTO_AHEAD=2 runs two per iteration)
    const long long n = (long long)groups * G, stride = 2 * (n + 4 * G);
    std::vector<float> hx((size_t)chains * stride, 0.0f);
    unsigned long long sd = 0x0A5EAD00ull;
    for (int ch = 0; ch < chains; ++ch)
        for (long long i = 0; i < n; ++i) {
            sd = sd * 6364136223846793005ull + 1442695040888963407ull;
            hx[(size_t)ch * stride + i] = (float)((double)(sd >> 40) / 8388608.0 - 1.0);
        }
    float *dx, *dy; unsigned long long *dt;
    hipMalloc(&dx, hx.size() * 4); hipMalloc(&dy, hx.size() * 4); hipMalloc(&dt, chains * 8);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dy, 0, hx.size() * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(chains), dim3(64), 0, 0, dx, dy, stride, groups, c, dt);
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed: %%s\n", hipGetErrorString(hipGetLastError())); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = ms < best ? ms : best;
    }
    std::vector<float> hy(hx.size());
    hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost);
    // parity, chains 0, 1 and 511: find the lag of the conveyor, then compare bit for bit
    int bad_total = 0;
    for (int ch : {0, 1, 511}) {
        std::vector<float> x(hx.begin() + (size_t)ch * stride, hx.begin() + (size_t)ch * stride + n), y;
        host_cascade(c, x, y);
        int lag = -1;
        for (int L = 0; L < 4 * G && lag < 0; ++L) {
            bool ok = true;
            for (int i = 0; i < 2000 && ok; ++i) ok = std::memcmp(&hy[(size_t)ch * stride + L + i], &y[i], 4) == 0 || (y[i] == 0.0f && hy[(size_t)ch * stride + L + i] == 0.0f);
            if (ok) lag = L;
        }
        long long bad = 0, cmp = 0;
        if (lag >= 0)
            for (long long i = 0; i + lag < n - G; ++i, ++cmp)
                bad += std::memcmp(&hy[(size_t)ch * stride + lag + i], &y[i], 4) != 0 && !(y[i] == 0.0f && hy[(size_t)ch * stride + lag + i] == 0.0f);
        printf("chain %%3d: lag %%d, %%lld of %%lld samples differ from the host's DF2T cascade%%s\n", ch, lag, bad, cmp, lag < 0 ? "  (NO LAG FITS)" : "");
        bad_total += lag < 0 || bad != 0;
    }
    printf("512 chains x %%lld samples, one chain per wave: %%.3f ms = %%.2f ns per sample  (k_eq_ring: 0.920 ms x 6 = 5.52 ms for 480 256 samples = 11.49 ns; %%d VALU per %%d samples here = %%.3f per sample against 6.5)\n",
           n, best, best * 1e6 / (double)n, %(n_valu)d, G, %(n_valu)d / (double)G);
    // a lone wave: shader-clock independent cycles via the 100 MHz counter is what s_memtime gives on this part; report ticks
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, stride, groups, c, dt);
    hipDeviceSynchronize();
    unsigned long long tk = 0; hipMemcpy(&tk, dt, 8, hipMemcpyDeviceToHost);
    printf("a lone wave: %%.3f s_memtime ticks per sample\n", (double)tk / (double)n);
    return bad_total ? 2 : 0;
}
''' % {"G": G, "asm": asm, "P": P.replace('"', '\\"'), "M": M.replace('"', '\\"'), "n_valu": n_valu}
    path = "/tmp/proto_eq_wave_ring.hip"
    open(path, "w").write(src)
    os.makedirs(os.path.join(HERE, "bin"), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-w", "-ffp-contract=off", "--offload-arch=gfx950", "-o",
                    os.path.join(HERE, "bin", "proto_eq_wave_ring" + ("" if VARIANT == "wave" else "_" + VARIANT) + ("_ahead2" if AHEAD == 2 else "")), path], check=True)
    print("built tools/bin/proto_eq_wave_ring (variant", VARIANT + ");", n_valu, "VALU per", G, "samples")


if __name__ == "__main__":
    main()
