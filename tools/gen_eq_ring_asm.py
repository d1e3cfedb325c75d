#!/usr/bin/env python3
"""Generates open_headstage_amd/csrc/experiments/eq_ring_v1_asm.inc: the hand-scheduled inline-asm body of
k_eq_ring's steady-state GROUP = 8 chunks x 16 systolic steps for the 4 chains of a wave.

The 16 lanes of a DPP row form a RING (see eq_kernels.hip).  Register X: lane L holds the sample that
band L filters next.  Band L's work is split over two lanes so that the hand-over to the next band
rides on the DPP operand of instructions that are needed anyway -- no move instruction at all:
    lane L   ("pre")  P  (b0x, b1x) = (b0, b1)[L] * X
    lane L+1 ("post") M  b2x = b2 * X(row_ror:1)          O  X' = b0x(row_ror:1) + s1   (= band L's output
                      T  t2 = b1x(row_ror:1) + s2              = band L+1's next input, already in place)
                      A  (a1,a2) * X'                      N  (s1,s2) = (t2, b2x) - ao
Lanes that hold no band pass samples on: (b0, b1) = (1, 0), everything else 0.  Every 4th step two more
instructions serve the I/O port: bank 3 (lanes 12..15) holds 4 finished samples -> CAP moves them into
one bank of the store register, INJ overwrites them with the next 4 inputs, which then rotate into lane
0 one per step.  6.5 VALU per sample (conveyor form, tools/gen_eq_asm.py: 9).

Registers: v102 X | v100 store register | v[106:107] (b0x, b1x) | v[108:109] (t2, b2x) |
v[110:111] (a1 out, a2 out) | v[112:113] (s1,s2).
Operands: %[s1] %[s2] %[o] (in/out), %[n0..n7] (out: next group's inputs, loads in flight on exit),
%[pb01] %[b2] %[a12], %[sp] store pointer, %[lp] prefetch pointer, %[x0..x7] inputs.
Lane 4b+i of an input / store register holds sample 16q + 4b + 3 - i (the pointers carry j ^ 3).
"""
import os
import sys

STEPS = 128
# tokens O T A N belong to step k, P M to step k+1; the I/O port goes in front of the first of P / M.
# Every dependent pair is two issue slots apart and every DPP read has two instructions between it and
# the VALU write of its source (gfx9 does not interlock that hazard).
ORDER = os.environ.get("EQ_RING_ORDER", "O T A P N M").split()
FULL = "row_mask:0xf bank_mask:0xf"
KNOCK = os.environ.get("EQ_RING_KNOCKOUT", "").split()      # timing experiments only: store / load / port
X = "v102"


def ror(n):
    n %= 16
    return "quad_perm:[0,1,2,3]" if n == 0 else f"row_ror:{n}"


def gen(first_step=0, loads=True):
    """first_step: 0 = whole group, 16 = chunks 1..7 only (the launch's first group, whose chunk 0 needs the
    gated form); loads: prefetch the next group's inputs inside the block (interior groups) or not (the
    caller has issued bounds-checked loads itself)."""
    out = []
    e = out.append

    P = "v_pk_mul_f32 v[106:107], %[pb01], v[102:103] op_sel_hi:[1,0]"
    M = f"v_mul_f32_dpp v109, {X}, %[b2] row_ror:1 {FULL}"
    O = f"v_add_f32_dpp {X}, v106, v112 row_ror:1 {FULL}"
    T = f"v_add_f32_dpp v108, v107, v113 row_ror:1 {FULL}"
    A = "v_pk_mul_f32 v[110:111], %[a12], v[102:103] op_sel_hi:[1,0]"
    N = "v_pk_add_f32 v[112:113], v[108:109], v[110:111] neg_lo:[0,1] neg_hi:[0,1]"

    def port(nxt):              # before step nxt (nxt % 4 == 0)
        if "port" in KNOCK:
            return
        b = (nxt // 4) % 4
        # bank 3 of X -> bank b of the store register; after b == 3: 16 finished samples of chunk nxt/16 - 1
        e(f"v_mov_b32_dpp v100, {X} {ror(4 * b + 4)} row_mask:0xf bank_mask:0x{1 << b:x}")
        if b == 3 and "store" not in KNOCK:
            q = nxt // 16
            e("global_store_dword %[sp], v100, off" + (f" offset:{64 * q}" if q else ""))
        # bank b of this chunk's input register -> bank 3 of X
        e(f"v_mov_b32_dpp {X}, %[x{nxt // 16}] {ror(12 - 4 * b)} row_mask:0xf bank_mask:0x8")

    e(".p2align 5")
    for q in range(8 if loads and "load" not in KNOCK else 0):
        e(f"global_load_dword %[n{q}], %[lp], off" + (f" offset:{64 * q}" if q else ""))
    e(f"v_mov_b32 {X}, %[o]")
    e("v_mov_b32 v112, %[s1]")
    e("v_mov_b32 v113, %[s2]")
    e("s_nop 1")
    port(first_step)
    e("s_nop 0")                # INJ -> M: a DPP read needs 2 wait states after the VALU write of its source
    e("s_nop 0")
    e(P)
    e(M)
    e("s_nop 0")                # P -> O likewise
    e("s_nop 0")
    for k in range(first_step, STEPS):
        last = k == STEPS - 1
        boundary = (k + 1) % 4 == 0 and not last
        ported = False
        for tok in (ORDER if not last else ["O", "T", "A", "N"]):
            if tok in ("P", "M") and boundary and not ported:
                port(k + 1)
                ported = True
            if tok == "NOP":
                e("v_nop_e64")      # 8 bytes: keeps the stream aligned (order experiments only)
            else:
                e({"O": O, "T": T, "A": A, "N": N, "P": P, "M": M}[tok])
    # A lone wave issues one instruction per ~4.1 cycles whatever its kind -- unless an 8-byte encoding
    # straddles a 32-byte fetch line (+~4 cycles each, tools/ubench_seq.py): keep every 8-byte
    # instruction of the stream 8-byte aligned, i.e. 4-byte ones (s_nop, v_mov_b32) only in pairs.
    off = 0
    for l in out[1:]:
        size = 4 if l.startswith(("s_nop", "v_mov_b32 ")) else 8
        assert size == 4 or off % 8 == 0, (l, off)
        off += size
    e("v_mov_b32 %[s1], v112")
    e("v_mov_b32 %[s2], v113")
    e(f"v_mov_b32 %[o], {X}")
    return out


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "open_headstage_amd/csrc/experiments/eq_ring_v1_asm.inc"
    name = sys.argv[2] if len(sys.argv) > 2 else "EQ_RING_BODY"
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_eq_ring_asm.py -- do not edit.  128-step EQ ring group bodies (see eq_kernels.hip).\n")
        for macro, kw in ((name, {}), (name + "_NOLOAD", {"loads": False}),
                          (name + "_HEAD", {"first_step": 16, "loads": False})):
            lines = gen(**kw)
            f.write(f"#define {macro} \\\n")
            f.write(" \\\n".join('    "' + l + '\\n"' for l in lines))
            f.write("\n\n")
    print("wrote", path)
