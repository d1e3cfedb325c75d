#!/usr/bin/env python3
"""Generates open_headstage_amd/csrc/eq_ring2_asm.inc: the hand-scheduled inline-asm bodies of the ring-form
EQ (k_eq_ring, eq_ring2_body.hpp) on LOW, PINNED registers -- 24 VGPRs for the whole steady state, so that an
EQ wave fits into the 32 registers per lane that four k_conv_p1 waves (4 x 120) leave free on a SIMD.

Arithmetic and issue order are those of round 2's tools/gen_eq_ring_asm.py (kept for the A/B, OHS_EQ_RING_V1=1):
the 16 lanes of a DPP row form a RING; band L's work is split over two lanes,
    lane L   ("pre")  P  (b0x, b1x) = (b0, b1)[L] * X
    lane L+1 ("post") M  b2x = b2 * X(row_ror:1)          O  X' = b0x(row_ror:1) + s1   (= band L's output
                      T  t2 = b1x(row_ror:1) + s2              = band L+1's next input, already in place)
                      A  (a1,a2) * X'                      N  (s1,s2) = (t2, b2x) - ao
6 VALU per step + the I/O port every 4th step (CAP: bank 3 of X -> one bank of the store register, INJ: the next
4 inputs -> bank 3 of X) = 6.5 VALU per sample; every 16th step the port also stores a finished output register and
refills the input register it has just emptied, the two memory instructions back to back.

What changed against round 2:
  * every register is pinned ("{vN}" constraints in eq_ring2_body.hpp); nothing is copied in or out of a block:
        v0  byte offset of this lane's sample slot (SGPR base + this, `saddr` addressing: no 64-bit address VGPRs)
        v1  b2          v2  X        v3  store register (also the unread upper half of the pair v[2:3])
        v[4:5] (b0x, b1x)   v[6:7] (t2, b2x)   v[8:9] (a1 out, a2 out)   v[10:11] (s1, s2)
        v[12:13] (b0, b1) of the pre lane   v[14:15] (a1, a2)   v16..v23 the group's 8 input registers
        v24 (NOLOAD forms only) chunk 0 of the NEXT group
        v25 second store register (interior and closing groups: the memory instructions of two chunks go out together,
            store, store, load, load -- a lone wave pays ~12 cycles for a vector-memory instruction between VALU
            instructions, ~17 for two back to back, ~20 for four: 11.82 -> 11.61 -> 11.43 ns per sample)
  * the input registers are reloaded IN PLACE: chunk q's register is dead after its last INJ (step 16q + 12) and
    the load of the next group's chunk q is issued right there -- 8 I/O registers instead of 16.  vmcnt is counted
    by hand (vector-memory operations retire in issue order): per group the queue sees S0 L0 S1 L1 ... S7 L7, and
    two waits per group, `vmcnt(8)` in front of chunk 4 and in front of the next group's chunk 0, retire the four
    loads those chunks consume.
  * the interior groups loop INSIDE the asm statement (counter in an SGPR), and every block is software-pipelined
    across its ends: on entry the first step's port + P + M have been executed (by the previous block or by the
    gated C++ form), on exit the next step's have.  A group costs 832 VALU + 16 VMEM + 6 other instructions
    (two waits, the offset advance, three loop instructions) -- the 4-byte ones in adjacent pairs, so that every
    8-byte encoding stays 8-byte aligned (a lone wave pays ~4 cycles for one that straddles a fetch line).

Blocks (first_step, loads, loop):
    HEAD_LOADS  16, yes, no    the launch's first group, whose chunk 0 ran in the gated form (the block loads chunks
                               2..7 of its own group as well)
    LOOP         0, yes, yes   interior groups: everything in bounds
    NOLOAD       0, no,  no    a full group whose successor is not full (the caller loaded v24)
    HEAD_NOLOAD 16, no,  no
Macro: EQ_RING2_MAIN = all of a launch's full groups in one statement -- HEAD_NOLOAD when there is one (%[only]),
else HEAD_LOADS + LOOP (%[cnt] >= 0 times) + NOLOAD.
Lane 4b+i of an input / store register holds sample 16q + 4b + 3 - i (the offsets carry j ^ 3).
"""
import os
import sys

STEPS = 128
ORDER = os.environ.get("EQ_RING_ORDER", "O T A P N M").split()
FULL = "row_mask:0xf bank_mask:0xf"
KNOCK = os.environ.get("EQ_RING_KNOCKOUT", "").split()      # timing experiments only: store / load / port / wait
WAITS = int(os.environ.get("EQ_RING2_WAITS", "2"))          # waits per group: 2 (default) or 4
WAITCNT = 8 if WAITS == 2 else 12                           # each wait retires the loads of the next 4 (2) chunks
# The port's store and load are issued BACK TO BACK, behind the port's second move (which reads the register the load
# refills): a lone wave pays ~12 cycles for a vector-memory instruction between VALU instructions but only ~4 more for a
# second one right behind it (tools/ubench_vmem_issue.hip) -- 11.82 -> 11.61 ns per sample.  EQ_RING2_VMEM_ADJ=0: round
# 3's first order (move, store, move, load).
VMEM_ADJ = os.environ.get("EQ_RING2_VMEM_ADJ", "1") == "1"
# Memory instructions of CLUSTER consecutive chunks issued together (1, 2 or 4): the outputs wait in CLUSTER store
# registers (v3, v25, v26, v27), the input registers of the cluster's chunks are refilled together.  Interior and closing
# groups only; the launch's first group keeps one store + one load per chunk.  The two waits per group stay where they
# are and keep their count (a group issues 16 memory instructions whatever their grouping).
CLUSTER = int(os.environ.get("EQ_RING2_CLUSTER", "2"))       # 1: 11.61, 2: 11.43, 4: 11.42-11.52 ns per sample
assert CLUSTER in (1, 2, 4)
assert not (CLUSTER == 4 and int(os.environ.get("EQ_RING2_WAITS", "2")) == 4), "four waits per group retire two chunks each: clusters of 1 or 2 chunks only"
VMEM_DELAY = int(os.environ.get("EQ_RING2_VMEM_DELAY", "0"))  # experiment: the port's store / load issued this many VALU
                                                            # instructions behind the DPP move that wrote the store register
VOFF, B2, X, ST = "v0", "v1", "v2", "v3"
XIN = [f"v{16 + q}" for q in range(8)]
XNEXT = "v24"
ST_REGS = [ST, "v25", "v26", "v27"]


def ror(n):
    n %= 16
    return "quad_perm:[0,1,2,3]" if n == 0 else f"row_ror:{n}"


def size_of(line):
    if line.startswith((".p2align", "1:", "2:", "3:", "4:")):
        return 0
    if line.startswith(("s_nop", "s_waitcnt", "s_sub_u32", "s_cmp", "s_cbranch", "s_branch", "v_add_u32_e32")):
        return 4
    return 8


def gen(first_step=0, loads=True, loop=False):
    out = []
    pending = []                # [countdown, instruction]: the port's VMEM instructions when VMEM_DELAY > 0

    def flush(force=False):
        while pending and (force or pending[0][0] <= 0):
            out.append(pending.pop(0)[1])

    def e(line):
        is_valu = line.startswith("v_") and not line.startswith("v_add_u32")
        if not is_valu:
            flush(force=True)   # waits, branches, the offset advance: nothing may be pending across them
        out.append(line)
        if is_valu:
            for it in pending:
                it[0] -= 1
            flush()

    def vmem(line):
        if VMEM_DELAY > 0:
            pending.append([VMEM_DELAY, line])
        else:
            out.append(line)

    P = f"v_pk_mul_f32 v[4:5], v[12:13], v[2:3] op_sel_hi:[1,0]"
    M = f"v_mul_f32_dpp v7, {X}, {B2} row_ror:1 {FULL}"
    O = f"v_add_f32_dpp {X}, v4, v10 row_ror:1 {FULL}"
    T = f"v_add_f32_dpp v6, v5, v11 row_ror:1 {FULL}"
    A = f"v_pk_mul_f32 v[8:9], v[14:15], v[2:3] op_sel_hi:[1,0]"
    N = f"v_pk_add_f32 v[10:11], v[6:7], v[8:9] neg_lo:[0,1] neg_hi:[0,1]"

    # Offsets: V = row offset + 4 (j ^ 3) + 512 g on entry of group g.  The LOOP form advances V by 512 in front of
    # chunk 4 (the instruction pairs with that wait); every other form leaves it alone.
    cl = CLUSTER if first_step == 0 else 1

    def store_imm(q, at=None):      # output chunk q - 1, issued at the port of chunk `at`
        adv = loop and (q if at is None else at) >= 4
        return 64 * q - 64 - (512 if adv else 0)

    def load_imm(q, at=None):
        adv = loop and (q if at is None else at) >= 4
        return 512 + 64 * q - (512 if adv else 0)

    def port(nxt):              # in front of step nxt (nxt % 4 == 0); nxt == 128: the next group's step 0
        if "port" in KNOCK:
            return
        b = (nxt // 4) % 4
        q = nxt // 16
        if loop and nxt == 64:
            # WAITS == 2: L4..L7 of the previous group (S0 L0 .. S3 L3 issued since); WAITS == 4: L4, L5 only
            e("s_nop 0" if "wait" in KNOCK else f"s_waitcnt vmcnt({WAITCNT})")
            e("v_add_u32_e32 v0, %[k512], v0")
        if head and b == 0 and 2 <= q <= 7:
            # chunk q's own register: issued since L'q: L'(q+1) .. L'7, [L0], S1 [L1] .. S(q-1) [L(q-1)]
            e(f"s_waitcnt vmcnt({6 + q if loads else 6})")
            e("s_nop 0")
        if loop and WAITS == 4 and nxt in (32, 96) and "wait" not in KNOCK:
            e(f"s_waitcnt vmcnt({WAITCNT})")
            e("s_nop 0")
        e(f"v_mov_b32_dpp {ST_REGS[q % cl]}, {X} {ror(4 * b + 4)} row_mask:0xf bank_mask:0x{1 << b:x}")
        cluster = b == 3 and (q + 1) % cl == 0
        if b == 3 and "store" not in KNOCK and not VMEM_ADJ:
            assert cl == 1
            vmem(f"global_store_dword {VOFF}, {ST}, %[dst] offset:{store_imm(q)}")
        if nxt == 128:
            src = XIN[0] if loads else XNEXT
        else:
            src = XIN[q]
        e(f"v_mov_b32_dpp {X}, {src} {ror(12 - 4 * b)} row_mask:0xf bank_mask:0x8")
        if cluster and "store" not in KNOCK and VMEM_ADJ:
            for qq in range(q - cl + 1, q + 1):
                vmem(f"global_store_dword {VOFF}, {ST_REGS[qq % cl]}, %[dst] offset:{store_imm(qq, q)}")
        if cluster and loads and "load" not in KNOCK:
            for qq in range(q - cl + 1, q + 1):
                vmem(f"global_load_dword {XIN[qq]}, {VOFF}, %[src] offset:{load_imm(qq, q)}")

    e(".p2align 5")
    head = first_step == 16
    if head:
        # Chunk 0 ran in the gated form, which only needed x0 and x1: the block requests chunks 2..7 of its OWN
        # group itself (six registers the gated code does not have to keep alive), then -- x0 is free from the
        # start -- the next group's chunk 0.  Queue: L'2 .. L'7 [L0] | S1 [L1] | S2 [L2] | ...
        for q in range(2, 8):
            e(f"global_load_dword {XIN[q]}, {VOFF}, %[src] offset:{64 * q}")
        if loads and "load" not in KNOCK:
            e(f"global_load_dword {XIN[0]}, {VOFF}, %[src] offset:{load_imm(0)}")
    # the compiler's code wrote v4, v5, v7 last: a DPP read needs 2 wait states after the VALU write of its source
    e("s_nop 1")
    e("s_nop 0")
    if loop:
        e("1:")
    for k in range(first_step, STEPS):
        last = k == STEPS - 1
        boundary = (k + 1) % 4 == 0
        ported = False
        for tok in ORDER:
            if tok in ("P", "M") and boundary and not ported:
                if last and loads:
                    # x0..x3 of the next group (S4 L4 .. S7 L7 issued since); pairs with the counter decrement
                    e("s_nop 0" if "wait" in KNOCK else f"s_waitcnt vmcnt({WAITCNT})")
                    e("s_sub_u32 %[cnt], %[cnt], 1" if loop else "s_nop 0")
                port(k + 1)
                ported = True
            if tok == "NOP":
                e("v_nop_e64")      # 8 bytes: keeps the stream aligned (order experiments only)
            else:
                e({"O": O, "T": T, "A": A, "N": N, "P": P, "M": M}[tok])
    flush(force=True)
    if loop:
        e("s_cmp_lg_u32 %[cnt], 0")
        e("s_cbranch_scc1 1b")
    if loads:
        e("s_waitcnt vmcnt(0)")     # the in-place loads of the last group: the registers are the caller's from here
        e("s_nop 1")                # (and the compiler may read v2 / v4.. through DPP right behind the block)
    else:
        e("s_nop 1")
        e("s_nop 0")
    check_alignment(out)
    return out


def check_alignment(lines):
    """every 8-byte encoding 8-byte aligned (blocks start at a .p2align 5)"""
    off = 0
    for l in lines:
        if l.startswith(".p2align"):
            off = 0
            continue
        size = size_of(l)
        assert size != 8 or off % 8 == 0, (l, off)
        off += size
    assert off % 8 == 0


def gen_main():
    """The launch's full groups, all of them, in ONE statement (one statement: two in an if / else made hipcc keep
    copies of the pinned state for the path not taken, beyond the kernel's 32 registers):
        %[only] != 0 (one full group):   HEAD_NOLOAD
        else (%[cnt] + 2 full groups):   HEAD_LOADS, the interior loop %[cnt] >= 0 times, NOLOAD
    v0 enters as the lane's offset of group 0 and leaves as that of the last full group."""
    out = [".p2align 5",
           "s_cmp_lg_u32 %[only], 0",
           "s_cbranch_scc1 3f"]
    out += gen(first_step=16)
    out += ["v_add_u32_e32 v0, %[k512], v0",       # group 1's offset: what the loop / the closing group expects
            "s_cmp_eq_u32 %[cnt], 0",
            "s_cbranch_scc1 2f",
            "s_nop 0"]
    out += gen(loop=True)
    out += ["2:"]
    out += gen(loads=False)
    out += ["s_branch 4f", "s_nop 0", "3:"]
    out += gen(first_step=16, loads=False)
    out += ["4:"]
    check_alignment(out)
    return out


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "open_headstage_amd/csrc/eq_ring2_asm.inc"
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_eq_ring2_asm.py -- do not edit.  EQ ring group bodies on pinned low registers "
                "(see eq_ring2_body.hpp).\n")
        for macro, lines in (("EQ_RING2_MAIN", gen_main()),):
            f.write(f"#define {macro} \\\n")
            f.write(" \\\n".join('    "' + l + '\\n"' for l in lines))
            f.write("\n\n")
    print("wrote", path)
