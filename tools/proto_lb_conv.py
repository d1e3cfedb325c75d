#!/usr/bin/env python3
"""CPU prototype of the block-2048 / FFT-4096 plan for long impulse responses (conv_lb_kernels.hip), index for index:

  forward  (k_conv_lb_forward): window [x_{u-1}, x_u] (4096 packed samples z = x_l + j x_r), decimation in frequency --
           a lane-local radix-4 step over the window's quarters, twiddle w4096^(n1 c), then four 1024-point transforms,
           wave c producing the bins k = 4 k1 + c;  stored in the PAIR layout: slot q holds (Z[q], Z[4096 - q]),
           slot 0 holds (Z[0], Z[2048]).
  product  (k_conv_lb_mac_inverse): W_t = sum_i Z_{t-i} C_i + conj(Z_{t-i}[N - k]) D_i on pairs, C = (A - jB)/2,
           D = (A + jB)/2, A = H_lsl + j H_lsr, B = H_rsl + j H_rsr (spectra of 2048-tap partitions, 1/4096 folded in).
  inverse  four 1024-point inverse transforms (wave c: bins 4 k1 + c), twiddle conj(w4096^(n1 c)), radix-4 step for the
           window's second half only (overlap-save): frames n1 + 1024 (n2 - 2), n2 = 2, 3.

Checked against direct convolution in f64.  Run: python tools/proto_lb_conv.py
"""
import numpy as np

N = 4096
B = 2048
Q = 1024


def forward_dif(z):
    """z: complex[4096] -> Z[4096] natural order, through the radix-4 + four FFT-1024 route"""
    n1 = np.arange(Q)
    quarters = z.reshape(4, Q)                       # quarters[n2][n1] = z[n1 + 1024 n2]
    Z = np.empty(N, complex)
    for c in range(4):
        u = sum(quarters[n2] * np.exp(-2j * np.pi * n2 * c / 4) for n2 in range(4))      # radix-4 over n2
        v = u * np.exp(-2j * np.pi * n1 * c / N)                                          # w4096^(n1 c)
        Z[c::4] = np.fft.fft(v)                                                           # bins 4 k1 + c
    return Z


def to_pairs(Z):
    """pair layout [2048][2]: slot q = (Z[q], Z[4096 - q]); slot 0 = (Z[0], Z[2048])"""
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
    """W: natural-order spectrum -> the window's second half (2048 complex samples), unnormalised"""
    n1 = np.arange(Q)
    y = [np.fft.ifft(W[c::4]) * Q * np.exp(2j * np.pi * n1 * c / N) for c in range(4)]   # unnormalised IFFT-1024, conj twiddle
    out = np.empty(B, complex)
    for n2 in (2, 3):
        out[(n2 - 2) * Q:(n2 - 1) * Q] = sum(y[c] * np.exp(2j * np.pi * n2 * c / 4) for c in range(4))
    return out


def pair_mac(acc, zp, cd):
    """acc[2048][2] += pair product; zp [2048][2]; cd [2048][4] = (C[k], D[k], C[N-k], D[N-k])"""
    za, zb = zp[:, 0].copy(), zp[:, 1].copy()
    ma, mb = zb.copy(), za.copy()
    ma[0], mb[0] = za[0], zb[0]                      # slot 0: bins 0 and 2048 mirror themselves
    acc[:, 0] += za * cd[:, 0] + np.conj(ma) * cd[:, 1]
    acc[:, 1] += zb * cd[:, 2] + np.conj(mb) * cd[:, 3]


def tables(irs, P2):
    """irs: four real impulse responses -> cd[P2][2048][4]"""
    L = P2 * B
    h = [np.pad(np.asarray(x, float), (0, L - len(x))) for x in irs]
    out = np.empty((P2, B, 4), complex)
    for i in range(P2):
        sl = slice(i * B, (i + 1) * B)
        A = forward_dif(np.concatenate([h[0][sl] + 1j * h[1][sl], np.zeros(B)]))          # H_lsl + j H_lsr
        Bs = forward_dif(np.concatenate([h[2][sl] + 1j * h[3][sl], np.zeros(B)]))         # H_rsl + j H_rsr
        C = (A - 1j * Bs) / 2 / N
        D = (A + 1j * Bs) / 2 / N
        Cp, Dp = to_pairs(C), to_pairs(D)
        out[i, :, 0], out[i, :, 1], out[i, :, 2], out[i, :, 3] = Cp[:, 0], Dp[:, 0], Cp[:, 1], Dp[:, 1]
    return out


def lb_convolve(xl, xr, irs, hist_l=None, hist_r=None):
    """xl, xr: the call's frames (any multiple of 512); hist_*: the frames before the call (>= P2 * 2048 of them, or
    None = zeros).  -> (out_l, out_r)"""
    taps = max(len(h) for h in irs)
    P2 = max(1, -(-taps // B))
    n = len(xl)
    nbig = -(-n // B)
    Lh = P2 * B
    hl = np.zeros(Lh) if hist_l is None else np.concatenate([np.zeros(Lh), hist_l])[-Lh:]
    hr = np.zeros(Lh) if hist_r is None else np.concatenate([np.zeros(Lh), hist_r])[-Lh:]
    zl = np.concatenate([hl, xl, np.zeros(nbig * B - n)])
    zr = np.concatenate([hr, xr, np.zeros(nbig * B - n)])
    z = zl + 1j * zr
    cd = tables(irs, P2)
    # ring: spectra of windows u = -(P2 - 1) .. nbig - 1; window u covers frames [(u - 1) B, (u + 1) B) of the call
    ring = {}
    for u in range(-(P2 - 1), nbig):
        a = Lh + (u - 1) * B
        ring[u] = to_pairs(forward_dif(z[a:a + N]))
    out = np.empty(nbig * B, complex)
    for t in range(nbig):
        acc = np.zeros((B, 2), complex)
        for i in range(P2):
            pair_mac(acc, ring[t - i], cd[i])
        out[t * B:(t + 1) * B] = inverse_dit_second_half(from_pairs(acc))
    return out.real[:n], out.imag[:n]


def main():
    rng = np.random.default_rng(1)
    z = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    assert np.allclose(forward_dif(z), np.fft.fft(z))
    assert np.allclose(from_pairs(to_pairs(z)), z)
    W = np.fft.fft(z)
    assert np.allclose(inverse_dit_second_half(W), z[B:] * N)
    for taps, n, pre in ((5000, 7 * 512, 0), (16384, 9 * 2048, 20000), (2049, 4096 + 512, 3000), (700, 1536, 5000)):
        irs = [rng.standard_normal(t) * np.exp(-np.arange(t) / (taps / 6)) for t in (taps, taps - 3, taps // 2, taps)]
        xl_all, xr_all = rng.uniform(-1, 1, pre + n), rng.uniform(-1, 1, pre + n)
        yl, yr = lb_convolve(xl_all[pre:], xr_all[pre:], irs, xl_all[:pre] if pre else None, xr_all[:pre] if pre else None)
        cv = lambda x, h: np.convolve(x, h)[pre:pre + n]
        ref_l = cv(xl_all, irs[0]) + cv(xr_all, irs[2])
        ref_r = cv(xl_all, irs[1]) + cv(xr_all, irs[3])
        err = np.sqrt(np.mean((yl - ref_l) ** 2 + (yr - ref_r) ** 2) / np.mean(ref_l ** 2 + ref_r ** 2))
        print(f"taps {taps:6d} frames {n:6d} history {pre:6d}: relative RMS error {err:.2e}")
        assert err < 1e-12
    print("ok")


if __name__ == "__main__":
    main()
