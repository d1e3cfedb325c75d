#!/usr/bin/env python3
"""CPU prototype of the block-8192 / FFT-16384 fused plan for long impulse responses (conv_xb_kernels.hip), index for index:

  one workgroup = a run of consecutive 8192-frame blocks of one stream; per block
  forward  window [x_{i-1}, x_i] (16384 packed samples z = x_l + j x_r), decimation in frequency: a lane-local radix-16 step over
           the window's sixteenths (two radix-4 stages, the way dft16() in the kernel does it), twiddle w16384^(n1 c), then sixteen
           1024-point transforms, wave c producing the bins k = 16 k1 + c;  exchanged into PAIRS: slot q = (Z[q], Z[16384 - q]),
           slot 0 = (Z[0], Z[8192])
  product  two partitions of 8192 taps at most, and no spectrum ever leaves the workgroup: W_i = Z_i CD_0 + carry, carry' = Z_i CD_1
           (the run's first window is transformed for its carry alone)
  inverse  sixteen 1024-point inverse transforms (wave c: bins 16 k1 + c), twiddle conj(w16384^(n1 c)), radix-16 step, of which only
           the window's second half is kept (overlap-save): frames n1 + 1024 (n2 - 8), n2 = 8 .. 15

Checked against direct convolution in f64.  Run: python tools/proto_xb_conv.py
"""
import numpy as np

N = 16384
B = 8192
Q = 1024
R = 16


def dft16(z, sign):
    """z[16] (arrays) -> X[16], X[c] = sum_n2 w16^(sign n2 c) z[n2], as two radix-4 stages:
    T_b[c1] = sum_a w4^(a c1) z[4a + b];  X[c1 + 4 c2] = sum_b w4^(b c2) (w16^(b c1) T_b[c1])"""
    w4 = lambda e: np.exp(sign * 2j * np.pi * e / 4)
    w16 = lambda e: np.exp(sign * 2j * np.pi * e / 16)
    T = [[sum(w4(a * c1) * z[4 * a + b] for a in range(4)) for c1 in range(4)] for b in range(4)]
    X = [None] * 16
    for c1 in range(4):
        for c2 in range(4):
            X[c1 + 4 * c2] = sum(w4(b * c2) * w16(b * c1) * T[b][c1] for b in range(4))
    return X


def forward_dif(z):
    n1 = np.arange(Q)
    parts = z.reshape(R, Q)                          # parts[n2][n1] = z[n1 + 1024 n2]
    u = dft16([parts[n2] for n2 in range(R)], -1)
    Z = np.empty(N, complex)
    for c in range(R):
        Z[c::R] = np.fft.fft(u[c] * np.exp(-2j * np.pi * n1 * c / N))       # bins 16 k1 + c
    return Z


def to_pairs(Z):
    P = np.empty((B, 2), complex)
    P[0] = (Z[0], Z[B])
    q = np.arange(1, B)
    P[1:, 0] = Z[q]
    P[1:, 1] = Z[N - q]
    return P


def from_pairs(P):
    Z = np.empty(N, complex)
    Z[0], Z[B] = P[0]
    q = np.arange(1, B)
    Z[q] = P[1:, 0]
    Z[N - q] = P[1:, 1]
    return Z


def inverse_dit_second_half(W):
    n1 = np.arange(Q)
    y = [np.fft.ifft(W[c::R]) * Q * np.exp(2j * np.pi * n1 * c / N) for c in range(R)]
    x = dft16(y, +1)
    return np.concatenate(x[8:])


def pair_mul(zp, cd):
    za, zb = zp[:, 0].copy(), zp[:, 1].copy()
    ma, mb = zb.copy(), za.copy()
    ma[0], mb[0] = za[0], zb[0]
    out = np.empty_like(zp)
    out[:, 0] = za * cd[:, 0] + np.conj(ma) * cd[:, 1]
    out[:, 1] = zb * cd[:, 2] + np.conj(mb) * cd[:, 3]
    return out


def tables(irs, P2):
    L = P2 * B
    h = [np.pad(np.asarray(x, float), (0, L - len(x))) for x in irs]
    out = np.empty((P2, B, 4), complex)
    for i in range(P2):
        sl = slice(i * B, (i + 1) * B)
        A = forward_dif(np.concatenate([h[0][sl] + 1j * h[1][sl], np.zeros(B)]))
        Bs = forward_dif(np.concatenate([h[2][sl] + 1j * h[3][sl], np.zeros(B)]))
        C = (A - 1j * Bs) / 2 / N
        D = (A + 1j * Bs) / 2 / N
        Cp, Dp = to_pairs(C), to_pairs(D)
        out[i, :, 0], out[i, :, 1], out[i, :, 2], out[i, :, 3] = Cp[:, 0], Dp[:, 0], Cp[:, 1], Dp[:, 1]
    return out


def xb_convolve(xl, xr, irs, hist_l=None, hist_r=None, run=3):
    taps = max(len(h) for h in irs)
    P2 = max(1, -(-taps // B))
    assert P2 <= 2
    n = len(xl)
    nblk = -(-n // B)
    Lh = 2 * B
    hl = np.zeros(Lh) if hist_l is None else np.concatenate([np.zeros(Lh), hist_l])[-Lh:]
    hr = np.zeros(Lh) if hist_r is None else np.concatenate([np.zeros(Lh), hist_r])[-Lh:]
    z = np.concatenate([hl, xl, np.zeros(nblk * B - n)]) + 1j * np.concatenate([hr, xr, np.zeros(nblk * B - n)])
    window = lambda i: z[Lh + (i - 1) * B:Lh + (i + 1) * B]      # block i of the call and the block in front of it
    cd = tables(irs, P2)
    out = np.empty(nblk * B, complex)
    for i0 in range(0, nblk, run):                                # one workgroup
        carry = None
        if P2 == 2:
            carry = pair_mul(to_pairs(forward_dif(window(i0 - 1))), cd[1])
        for i in range(i0, min(i0 + run, nblk)):
            Z = to_pairs(forward_dif(window(i)))
            acc = pair_mul(Z, cd[0])
            if P2 == 2:
                acc += carry
                carry = pair_mul(Z, cd[1])
            out[i * B:(i + 1) * B] = inverse_dit_second_half(from_pairs(acc))
    return out.real[:n], out.imag[:n]


def main():
    rng = np.random.default_rng(1)
    z = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    x16 = [rng.standard_normal(3) + 1j * rng.standard_normal(3) for _ in range(16)]
    ref16 = np.fft.fft(np.array(x16), axis=0)
    assert np.allclose(np.array(dft16(x16, -1)), ref16)
    assert np.allclose(forward_dif(z), np.fft.fft(z))
    assert np.allclose(inverse_dit_second_half(np.fft.fft(z)), z[B:] * N)
    for taps, n, pre in ((16384, 9 * 8192 + 512, 40000), (9000, 5 * 8192, 0), (8192, 3 * 8192 + 1536, 3000), (6200, 2 * 8192, 20000)):
        irs = [rng.standard_normal(t) * np.exp(-np.arange(t) / (taps / 6)) for t in (taps, taps - 3, taps // 2, taps)]
        xl_all, xr_all = rng.uniform(-1, 1, pre + n), rng.uniform(-1, 1, pre + n)
        yl, yr = xb_convolve(xl_all[pre:], xr_all[pre:], irs, xl_all[:pre] if pre else None, xr_all[:pre] if pre else None)
        cv = lambda x, h: np.convolve(x, h)[pre:pre + n]
        ref_l = cv(xl_all, irs[0]) + cv(xr_all, irs[2])
        ref_r = cv(xl_all, irs[1]) + cv(xr_all, irs[3])
        err = np.sqrt(np.mean((yl - ref_l) ** 2 + (yr - ref_r) ** 2) / np.mean(ref_l ** 2 + ref_r ** 2))
        print(f"taps {taps:6d} frames {n:6d} history {pre:6d}: relative RMS error {err:.2e}")
        assert err < 1e-12
    print("ok")


if __name__ == "__main__":
    main()
