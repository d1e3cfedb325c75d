"""numpy prototype of the per-wave FFT-1024 index plan used by csrc/wave_fft.hpp.

One wavefront (64 lanes) x 16 complex registers per lane = 1024 points.
Forward:  time layout  lane m, reg a  : z[64a+m]
          A: DFT16 over a -> p ; twiddle w1024^(m p) ; T1 (LDS) ;
          B: DFT16 over b -> q ; twiddle w64^(c q)   ; T2 (LDS) ;
          C: DFT4  over c -> r
          bin layout   lane L=4q+p_hi, reg 4r+j   : Z[4L + j + 256 r]
Inverse runs the same steps backwards with conjugated twiddles.
Run: python tools/proto_wave_fft.py   (asserts, prints max errors)
"""
import numpy as np

N = 1024
def w(n, e):  # exp(-2 pi i e / n)
    return np.exp(-2j * np.pi * e / n)

def addr1(p, m):            # T1 image, complex units
    return 68 * p + m
def sigma(q, p_hi):
    return 16 * (p_hi & 1) + 8 * (p_hi >> 1) + (q & 7) + 32 * (q >> 3)
def addr2(p_lo, c, q, p_hi):  # T2 image
    return 65 * (4 * p_lo + c) + sigma(q, p_hi)

def dft_small(v, n, sign):
    k = np.arange(n)
    M = np.exp(sign * -2j * np.pi * np.outer(k, k) / n)
    return M @ v

def fwd(z):
    regs = np.zeros((64, 16), complex)
    for m in range(64):
        for a in range(16):
            regs[m, a] = z[64 * a + m]
    # A + twiddle 1
    for m in range(64):
        u = dft_small(regs[m], 16, +1)
        for p in range(16):
            u[p] *= w(1024, m * p)
        regs[m] = u
    lds = np.zeros(16 * 68, complex)
    for m in range(64):
        for p in range(16):
            lds[addr1(p, m)] = regs[m, p]
    for L1 in range(64):
        p, c = L1 >> 2, L1 & 3
        for b in range(16):
            regs[L1, b] = lds[addr1(p, 4 * b + c)]
    # B + twiddle 2
    for L1 in range(64):
        c = L1 & 3
        u = dft_small(regs[L1], 16, +1)
        for q in range(16):
            u[q] *= w(64, c * q)
        regs[L1] = u
    lds = np.zeros(16 * 65 + 64, complex)
    for L1 in range(64):
        p, c = L1 >> 2, L1 & 3
        p_hi, p_lo = p >> 2, p & 3
        for q in range(16):
            lds[addr2(p_lo, c, q, p_hi)] = regs[L1, q]
    for L2 in range(64):
        q, p_hi = L2 >> 2, L2 & 3
        for p_lo in range(4):
            for c in range(4):
                regs[L2, 4 * p_lo + c] = lds[addr2(p_lo, c, q, p_hi)]
    # C: DFT4 over c for each p_lo; result stored reg 4r + j (j = p_lo)
    out = np.zeros((64, 16), complex)
    for L2 in range(64):
        for j in range(4):
            u = dft_small(regs[L2, 4 * j:4 * j + 4], 4, +1)
            for r in range(4):
                out[L2, 4 * r + j] = u[r]
    return out

def bins_of(L, reg):
    r, j = reg >> 2, reg & 3
    return 4 * L + j + 256 * r

def inv(W):  # W in bin layout -> time layout (unnormalised)
    regs = np.zeros((64, 16), complex)
    for L2 in range(64):
        for j in range(4):
            v = np.array([W[L2, 4 * r + j] for r in range(4)])
            u = dft_small(v, 4, -1)               # over r -> c
            q = L2 >> 2
            for c in range(4):
                regs[L2, 4 * j + c] = u[c] * np.conj(w(64, c * q))
    lds = np.zeros(16 * 65 + 64, complex)
    for L2 in range(64):
        q, p_hi = L2 >> 2, L2 & 3
        for p_lo in range(4):
            for c in range(4):
                lds[addr2(p_lo, c, q, p_hi)] = regs[L2, 4 * p_lo + c]
    for L1 in range(64):
        p, c = L1 >> 2, L1 & 3
        for q in range(16):
            regs[L1, q] = lds[addr2(p & 3, c, q, p >> 2)]
    for L1 in range(64):
        p, c = L1 >> 2, L1 & 3
        u = dft_small(regs[L1], 16, -1)           # over q -> b
        for b in range(16):
            u[b] *= np.conj(w(1024, (4 * b + c) * p))
        regs[L1] = u
    lds = np.zeros(16 * 68, complex)
    for L1 in range(64):
        p, c = L1 >> 2, L1 & 3
        for b in range(16):
            lds[addr1(p, 4 * b + c)] = regs[L1, b]
    for m in range(64):
        for p in range(16):
            regs[m, p] = lds[addr1(p, m)]
    out = np.zeros((64, 16), complex)
    for m in range(64):
        out[m] = dft_small(regs[m], 16, -1)       # over p -> a
    return out

def mirror_src(L, reg):
    """(lane, reg) holding bin N-k for the bin k held at (L, reg)."""
    r, j = reg >> 2, reg & 3
    if j:
        return 63 - L, 4 * (3 - r) + (4 - j)
    if L:
        return 64 - L, 4 * (3 - r)
    return 0, 4 * ((4 - r) & 3)

if __name__ == "__main__":
    rng = np.random.default_rng(1)
    z = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    Z = fwd(z)
    ref = np.fft.fft(z)
    err = max(abs(Z[L, g] - ref[bins_of(L, g)]) for L in range(64) for g in range(16))
    print("fwd max err", err); assert err < 1e-9
    for L in range(64):
        for g in range(16):
            sl, sg = mirror_src(L, g)
            assert bins_of(sl, sg) == (N - bins_of(L, g)) % N
    y = inv(Z)
    err = max(abs(y[m, a] / N - z[64 * a + m]) for m in range(64) for a in range(16))
    print("roundtrip max err", err); assert err < 1e-9
    # LDS bank-conflict census (b64 accesses; complex address = bank pair)
    def worst(groups, f, nb):
        wv = 0
        for g in groups:
            for reg in range(16):
                cnt = {}
                for lane in g:
                    bk = f(lane, reg) % nb
                    cnt[bk] = cnt.get(bk, 0) + 1
                wv = max(wv, max(cnt.values()))
        return wv
    halves = [range(0, 32), range(32, 64)]
    g16 = [range(i, i + 16) for i in range(0, 64, 16)]
    print("T1 fwd write", worst(g16, lambda m, p: addr1(p, m), 16),
          "read", worst(halves, lambda L, b: addr1(L >> 2, 4 * b + (L & 3)), 32))
    print("T2 fwd write", worst(g16, lambda L, q: addr2((L >> 2) & 3, L & 3, q, L >> 4), 16),
          "read", worst(halves, lambda L, g: addr2(g >> 2, g & 3, L >> 2, L & 3), 32))
    print("T2 inv write", worst(g16, lambda L, g: addr2(g >> 2, g & 3, L >> 2, L & 3), 16),
          "read", worst(halves, lambda L, q: addr2((L >> 2) & 3, L & 3, q, L >> 4), 32))
    print("T1 inv write", worst(g16, lambda L, b: addr1(L >> 2, 4 * b + (L & 3)), 16),
          "read", worst(halves, lambda m, p: addr1(p, m), 32))
    # round 2: the mirror spectrum is read from the T2 image (wave_fft_fwd_mirror) -- groups j' = 1..3 of lane
    # 63 - L, group j' = 0 of lane 64 - L -- and one twiddle table w1024^(m p) in the T1 layout serves both
    # directions (rows forward, columns inverse)
    print("T2 mirror reads", worst(halves, lambda L, g: addr2(g >> 2, g & 3, (63 - L) >> 2, (63 - L) & 3), 32),
          worst(halves, lambda L, g: addr2(0, g & 3, ((64 - L) & 63) >> 2, ((64 - L) & 63) & 3), 32))
    print("twiddle table rows / columns", worst(halves, lambda m, p: addr1(p, m), 32),
          worst(halves, lambda L, b: addr1(L >> 2, 4 * b + (L & 3)), 32))
    # w64^(c q) lives in the four padding slots of row q of that table (tabB_index): distinct ADDRESSES per bank pair
    # (lanes that read the same address are one broadcast access)
    def worst_distinct(groups, f, nb):
        wv = 0
        for g in groups:
            for reg in range(16):
                per = {}
                for lane in g:
                    a = f(lane, reg)
                    per.setdefault(a % nb, set()).add(a)
                wv = max(wv, max(len(v) for v in per.values()))
        return wv
    tab_b = lambda q, c: 68 * q + 64 + c
    assert all(tab_b(q, c) < 16 * 68 and tab_b(q, c) % 68 >= 64 for q in range(16) for c in range(4))
    print("w64 table in the row padding, forward / inverse", worst_distinct(halves, lambda L, q: tab_b(q, L & 3), 32),
          worst_distinct(halves, lambda L, c: tab_b(L >> 2, c & 3), 32))
    # a wave's slab: 15 * 68 + 64 entries hold both images
    slab = 15 * 68 + 64
    assert max(addr1(p, m) for p in range(16) for m in range(64)) < slab
    assert max(addr2(a, b, c, d) for a in range(4) for b in range(4) for c in range(16) for d in range(4)) < slab
    print("slab", slab, "entries: OK")
    # the mirror trick itself: DFT4 over c of the T2-image groups of the mirror lanes gives Z[N - k]
    def t2_image(z):
        regs = np.zeros((64, 16), complex)
        for m in range(64):
            u = dft_small(np.array([z[64 * a + m] for a in range(16)]), 16, +1)
            for pp in range(16):
                regs[m, pp] = u[pp] * w(1024, m * pp)
        lds = np.zeros(16 * 68, complex)
        for m in range(64):
            for pp in range(16):
                lds[addr1(pp, m)] = regs[m, pp]
        for L1 in range(64):
            pp, c = L1 >> 2, L1 & 3
            u = dft_small(np.array([lds[addr1(pp, 4 * b + c)] for b in range(16)]), 16, +1)
            for q in range(16):
                regs[L1, q] = u[q] * w(64, c * q)
        img = np.zeros(16 * 65 + 64, complex)
        for L1 in range(64):
            pp, c = L1 >> 2, L1 & 3
            for q in range(16):
                img[addr2(pp & 3, c, q, pp >> 2)] = regs[L1, q]
        return img
    img = t2_image(z)
    for L in range(64):
        la, lb = 63 - L, (64 - L) & 63
        m = np.zeros(16, complex)
        for jj in range(4):
            src = lb if jj == 0 else la
            m[4 * jj:4 * jj + 4] = dft_small(np.array([img[addr2(jj, c, src >> 2, src & 3)] for c in range(4)]), 4, +1)
        for r in range(4):
            for j in range(4):
                got = m[4 * (4 - j) + (3 - r)] if j else (m[(4 - r) & 3] if L == 0 else m[3 - r])
                assert abs(got - ref[(N - bins_of(L, 4 * r + j)) % N]) < 1e-9, (L, r, j)
    print("mirror from the T2 image: OK")
    # round 3: the paired bin layout of k_conv_p1 (wave_fft.hpp: paired_group) -- bins k and N - k in one lane
    def paired_group(L, g):
        if g == 0:
            return L, 1
        if g == 1:
            return 63 - L, 3
        if L < 32:
            return (L if g == 2 else 63 - L), 2
        return (L if g == 2 else (64 - L) & 31), 0
    seen = set()
    for L in range(64):
        for g in range(4):
            seen.add(paired_group(L, g))
    assert len(seen) == 256 and seen == {(l, j) for l in range(64) for j in range(4)}      # every DFT4 group exactly once
    def paired_bin(L, g, r):
        lg, j = paired_group(L, g)
        return 4 * lg + j + 256 * r
    def paired_mirror(L, g, r):          # (slot, register) of the SAME lane holding bin N - k
        if g == 0:
            return 1, 3 - r
        if g == 1:
            return 0, 3 - r
        if L == 32:
            return (2, 3 - r) if g == 2 else (3, (4 - r) & 3)
        return (3 if g == 2 else 2), 3 - r
    for L in range(64):
        for g in range(4):
            for r in range(4):
                mg, mr = paired_mirror(L, g, r)
                assert paired_bin(L, mg, mr) == (N - paired_bin(L, g, r)) % N, (L, g, r)
    # numerics: DFT4 of the T2-image group gives the paired bins
    for L in range(64):
        for g in range(4):
            lg, j = paired_group(L, g)
            u = dft_small(np.array([img[addr2(j, c, lg >> 2, lg & 3)] for c in range(4)]), 4, +1)
            for r in range(4):
                assert abs(u[r] - ref[paired_bin(L, g, r)]) < 1e-9
    # bank census: register 4 g + c of lane L is element c of its slot-g group
    def paired_addr(L, reg):
        lg, j = paired_group(L, reg >> 2)
        return addr2(j, reg & 3, lg >> 2, lg & 3)
    print("T2 paired read", worst(halves, paired_addr, 32), "write", worst(g16, paired_addr, 16))
    # w64 rows of the inverse transform: slots 0 and 2 are groups of the lane itself (the same row)
    for L in range(64):
        assert paired_group(L, 0)[0] >> 2 == paired_group(L, 2)[0] >> 2 == L >> 2
    print("w64 rows, paired inverse", worst_distinct(halves, lambda L, reg: tab_b(paired_group(L, reg >> 2)[0] >> 2, reg & 3), 32))
    print("paired layout: OK")
    print("OK")
