#!/usr/bin/env python3
"""Where does a kernel's hot loop spill, and what is alive there?  Works on hipcc's assembly (no GPU needed).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 <the unit's flags> -S --cuda-device-only \\
          -Iinclude -Iopen_headstage_amd/csrc -o /tmp/k.s open_headstage_amd/csrc/conv_os_kernels.hip
    python tools/isa_liveness.py /tmp/k.s k_conv_p1_os [min_loop_instructions=2000]

Takes the smallest loop of the kernel with at least that many instructions (the steady state of a hand-unrolled hop /
block loop is one basic block), prints its instruction mix, a timeline (r / w LDS read / write, G / T global load / store,
S / L scratch store / load, <n> a run of n other instructions) and a backward-dataflow estimate of the VGPRs alive per
60-instruction window, then who is alive at the peak (distance to the defining and to the next using instruction).
This is the script behind DESIGN 4.3: it showed hipcc spilling LDS-loaded table values 300 instructions ahead of their
use because of a register peak at the FAR end of their live ranges."""
import collections
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        out += list(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else [int(m.group(3))]
    return out


def main():
    fn, kname = sys.argv[1], sys.argv[2]
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    lines = open(fn).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + kname + r"\w*:", l))
    end = next(i for i, l in enumerate(lines) if i > start and ".Lfunc_end" in l)
    lines = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and min_len < i - labels[m.group(1)]:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    if not loops:
        sys.exit(f"no loop of more than {min_len} lines in {kname}")
    # (prefer loops that do vector work: a kernel's launch-plan arithmetic can be a long scalar loop of its own)
    busy = [t for t in loops if sum(1 for x in lines[t[1]:t[2]] if x.strip().startswith(("ds_", "v_pk_", "v_fma", "v_add_f32"))) > 200]
    _, a, b = min(busy or loops)
    body = [x.strip().split(";")[0].strip() for x in lines[a:b + 1] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
    mix = collections.Counter()
    for t in body:
        op = t.split()[0]
        mix["packed" if op.startswith("v_pk") else "lds" if op.startswith("ds_") else "scratch" if op.startswith("scratch")
            else "vmem" if op.startswith(("global", "buffer", "flat")) else "salu" if op.startswith("s_") else "valu"] += 1
    print(f"{kname}: loop of {len(body)} instructions: {dict(mix)}")
    tl = "".join("S" if t.startswith("scratch_store") else "L" if t.startswith("scratch_load") else "w" if t.startswith("ds_write")
                 else "r" if t.startswith("ds_read") else "G" if t.startswith("global_load") else "T" if t.startswith("global_store")
                 else "." for t in body)
    print(re.sub(r"\.{20,}", lambda m: f"<{len(m.group(0))}>", tl))
    ins = []
    for t in body:
        parts = t.split(None, 1)
        op, toks = parts[0], [x.strip() for x in (parts[1] if len(parts) > 1 else "").split(",")]
        d, u = [], []
        if op.startswith(("global_store", "scratch_store", "ds_write", "flat_store", "buffer_store", "s_", "v_cmp")):
            for x in toks:
                u += regs(x)
        elif toks:
            d = regs(toks[0])
            for x in toks[1:]:
                u += regs(x)
            if op.startswith(("v_fmac", "v_mac", "v_pk_fmac")):
                u += d
        ins.append((op, set(d), set(u), t))
    live, sets = set(), [None] * len(ins)
    for _ in range(2):                      # twice: values carried around the loop
        for i in range(len(ins) - 1, -1, -1):
            live = (live - ins[i][1]) | ins[i][2]
            sets[i] = set(live)
    lv = [len(s) for s in sets]
    pk = lv.index(max(lv))
    print(f"VGPRs alive (values in registers: spilled ones do not count): peak {max(lv)} at instruction {pk}")
    for i in range(0, len(ins), 60):
        print(f"  {i:5d} {max(lv[i:i + 60]):4d}")

    def dist(r, i, field, step):
        for k in range(1, len(ins) + 1):
            j = (i + step * k) % len(ins)
            if r in ins[j][field]:
                return k
            if step > 0 and r in ins[j][1]:
                return None
        return None
    rows = [(r, dist(r, pk, 1, -1), dist(r, pk, 2, +1)) for r in sorted(sets[pk])]
    print("alive at the peak (register, instructions since its definition, instructions until its next use):")
    print("  " + " ".join(f"v{r}:{d}/{u}" for r, d, u in rows))


if __name__ == "__main__":
    main()
