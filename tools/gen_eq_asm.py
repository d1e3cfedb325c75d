#!/usr/bin/env python3
"""Generates open_headstage_amd/csrc/eq_group_asm.inc: the hand-scheduled inline-asm body of
k_eq_pass's steady-state GROUP = 8 chunks x 16 systolic steps for the 4 chains of a wave.

Per step 9 VALU in two 3-deep dependency chains (see eq_kernels.hip):
    O  out = s1 + b0x            T  t2 = s2 + b1x           Y  lane15 <- out[lane NB-1]   (row_shr:SHR)
    A  (a1,a2)*out               C  conveyor <<= 1 lane     D  x = row_shr:1(out), lane 0 <- conveyor
    N  (s1,s2) = (t2,b2x) - ao   M  b0*x                    P  (b1,b2)*x
Every step is exactly 64 bytes of code; the block starts on a 32-byte boundary (.p2align 5) because a
start at 28 mod 32 costs ~4 cycles per step (tools/ubench_eq_align.hip).  At each chunk boundary the
conveyor register (16 finished samples per row) is stored and refilled with the next 16 inputs.

Registers: v100 out | v[108:109] (b1x->t2, b2x) | v[110:111] (a1 out, a2 out) | v[112:113] (s1,s2) |
v114 b0x | v102/v104/v106 rotate through conveyor -> x -> scratch.
Operands: %[s1] %[s2] %[o] %[c] (in/out), %[n0..n7] (out: next group's inputs, loads in flight on exit),
%[b0] %[b12] %[a12], %[sp] store pointer, %[lp] prefetch pointer, %[x0..x7] inputs,
%[shr] (only in the SHR != 0 variant).

(An LDS-fed variant -- 7 VALU + 2 DS per step -- was measured at 60.8 cycles per step against 43.5 for
this one: a lone wave pays ~16 cycles per DS instruction.  Rejected.)
"""
import os
import sys

R = ["v102", "v104", "v106"]
RP = ["v[102:103]", "v[104:105]", "v[106:107]"]
STEPS = 128
ORDER = os.environ.get("EQ_ORDER", "O A T Y C D N M P").split()   # 39.7 cycles per step; "O T NOP Y A C D N M P" 43.6 (tools/ubench_eq_order.hip)


def gen(shr_nonzero: bool):
    out = []
    e = out.append

    def Y(i):
        y = R[(i + 1) % 3]
        e(f"v_mov_b32_dpp {y}, v100 row_shr:%[shr] row_mask:0xf bank_mask:0xf" if shr_nonzero else f"v_mov_b32 {y}, v100")

    def C(i):
        e(f"v_mov_b32_dpp {R[(i + 1) % 3]}, {R[i % 3]} row_shl:1 row_mask:0xf bank_mask:0xf")

    def D(i):
        e(f"v_mov_b32_dpp {R[i % 3]}, v100 row_shr:1 row_mask:0xf bank_mask:0xf")

    def M(i):
        e(f"v_mul_f32 v114, %[b0], {R[i % 3]}")

    def P(i):
        e(f"v_pk_mul_f32 v[108:109], %[b12], {RP[i % 3]} op_sel_hi:[1,0]")

    O = "v_add_f32 v100, v112, v114"
    T = "v_add_f32 v108, v113, v108"
    A = "v_pk_mul_f32 v[110:111], %[a12], v[100:101] op_sel_hi:[1,0]"
    N = "v_pk_add_f32 v[112:113], v[108:109], v[110:111] neg_lo:[0,1] neg_hi:[0,1]"

    def drain(i):          # conveyor of step i (= first step of chunk i/16) holds 16 finished samples
        e(f"global_store_dword %[sp], {R[i % 3]}, off offset:{64 * (i // 16)}")

    def refill(i):
        e(f"v_mov_b32 {R[i % 3]}, %[x{i // 16}]")

    e(".p2align 5")
    # prefetch of the NEXT group's 8 x 16 inputs per row: issued here, inside the block, so that the
    # compiler cannot place its conservative `s_waitcnt vmcnt(0)` between the loads and the compute
    for q in range(8):
        e(f"global_load_dword %[n{q}], %[lp], off" + (f" offset:{64 * q}" if q else ""))
    e("v_mov_b32 v100, %[o]")
    e("v_mov_b32 v112, %[s1]")
    e("v_mov_b32 v113, %[s2]")
    e(f"v_mov_b32 {R[0]}, %[c]")
    # chunk 0 boundary + front half of step 0
    drain(0)
    e("s_nop 1")
    refill(0)
    Y(0)
    e("s_nop 1")            # a DPP read needs 2 wait states after the VALU write of its source (refill)
    C(0); D(0); M(0); P(0)
    for k in range(STEPS):
        i = k + 1
        boundary = i < STEPS and i % 16 == 0
        if i < STEPS:
            # per-step issue order (tokens O T A N belong to step k, Y C D M P to step k+1); constraints:
            # Y needs 2 instructions after O (DPP read of v100), C after Y, D after C, M/P after D, N after A and T
            for tok in ORDER:
                if tok == "O": e(O)
                elif tok == "T": e(T)
                elif tok == "A": e(A)
                elif tok == "N": e(N)
                elif tok == "NOP": e("s_nop 0")
                elif tok == "Y":
                    Y(i)
                    if boundary:
                        refill(i)       # the conveyor was drained several instructions ago (below)
                elif tok == "C":
                    if boundary:
                        e("s_nop 3")    # refill -> C(i): DPP read needs wait states after the VALU write
                    C(i)
                elif tok == "D": D(i)
                elif tok == "M": M(i)
                elif tok == "P": P(i)
            if (i + 1) < STEPS and (i + 1) % 16 == 0:
                drain(i + 1)        # c_(i+1) = R[(i+1)%3] was completed by C(i) four instructions ago and is
                                    # refilled four instructions from now: clear of both ends
        else:
            e(O)
            e(T)
            e(A)
            e(N)
    cfinal = R[STEPS % 3]
    e("v_mov_b32 %[s1], v112")
    e("v_mov_b32 %[s2], v113")
    e("v_mov_b32 %[o], v100")
    e(f"v_mov_b32 %[c], {cfinal}")
    return out


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "open_headstage_amd/csrc/eq_group_asm.inc"
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_eq_asm.py -- do not edit.  128-step EQ group body (see eq_kernels.hip).\n")
        for name, flag in (("EQ_GROUP_BODY_SHR", True), ("EQ_GROUP_BODY_NOSHR", False)):
            lines = gen(flag)
            f.write(f"#define {name} \\\n")
            f.write(" \\\n".join('    "' + l + '\\n"' for l in lines))
            f.write("\n\n")
    print("wrote", path)
