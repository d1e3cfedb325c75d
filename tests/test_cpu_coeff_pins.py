"""Independent pins for the coefficient convenience path (ohs_biquad_coeffs / ohs_eq_update_band; SURVEY section 8 row
a12: biquad 0.4.2 Coefficients::<f32>::from_params on libm 0.1.4's sinf / cosf / powf).  The crates are not on this
machine, so their last bit stays PARITY UNPINNED; what CAN be checked without them, and without comparing a file with its
own copy:

  1. the musl restatement (csrc/libm_musl.h) against CORRECTLY ROUNDED results -- f64 evaluations rounded to f32 -- over
     more than 10^6 arguments of the domain the coefficient code can ask for (omega = 2 pi fc / fs for the CLAP range
     fc 20 ... 20 000 Hz at the common rates; 10^(dB / 40) for |dB| <= 16 and beyond): never more than ONE ulp away;
  2. the formulas and their operation order against an independent float32 re-implementation written here with numpy
     (each numpy f32 operation is one IEEE operation) over a sweep of the CLAP ranges (src/lib.rs:62-99: fc 20 ...
     20 000, Q 0.1 ... 10, gain +-16 dB, all 8 filter types): BIT-IDENTICAL when both use the same sin / cos / pow;
  3. the same sweep against the RBJ cookbook evaluated in f64: 93 % of the coefficients within 2 ulp; where one ulp of
     cos / sin / A moves the exact value by at most one ulp (well conditioned) within 2.5 ulp (shelves: 4, they take
     eight f32 operations), and elsewhere within what that ONE ulp of freedom can move the f64 result
     -- the low-pass b0 at 20 Hz is (1 - cos w) / 2 with cos w = 1 - 3.4e-6: one ulp of cos w is 2 % of it, in the
     crate's f32 arithmetic exactly as here.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = os.path.join(ROOT, "open_headstage_amd", "csrc", "libm_musl.h")
f32 = np.float32

SHIM = r'''
#include "%s"
void v_sinf(const float *x, float *y, long n) { for (long i = 0; i < n; ++i) y[i] = ohs_musl_sinf(x[i]); }
void v_cosf(const float *x, float *y, long n) { for (long i = 0; i < n; ++i) y[i] = ohs_musl_cosf(x[i]); }
void v_pow10(const float *x, float *y, long n) { for (long i = 0; i < n; ++i) y[i] = ohs_musl_powf(10.0f, x[i]); }
'''
RATES = (44100.0, 48000.0, 88200.0, 96000.0, 192000.0)


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    d = tmp_path_factory.mktemp("coeffpins")
    src = d / "shim.c"
    src.write_text(SHIM % PRODUCT)
    so = d / "shim.so"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c11", "-o", str(so), str(src), "-lm"],
                   check=True)
    return C.CDLL(str(so))


def _apply(lib, name, x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    getattr(lib, name)(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_long(x.size))
    return y


def _ulps(a, b):
    ia = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def _omega_f32(fc, fs):
    """2.0f * pi * fc / fs, left to right in f32 -- the argument the crate hands to sin / cos"""
    return (f32(2.0) * f32(3.14159265358979323846264338327950288) * np.asarray(fc, f32)) / np.asarray(fs, f32)


# ------------------------------------------------------------------------------------------------------------------
def test_musl_restatement_within_one_ulp_of_correctly_rounded_over_the_coefficient_domain(shim):
    rng = np.random.default_rng(2026)
    # omega: the CLAP range at five rates, log-uniform in fc (3 x 10^5) + every float32 between two bounds in steps
    fc = np.exp(rng.uniform(np.log(20.0), np.log(20000.0), 300_000)).astype(f32)
    om = np.concatenate([_omega_f32(fc, f32(fs)) for fs in RATES])
    lo, hi = f32(2.0 * np.pi * 20.0 / 192000.0).view(np.int32), f32(2.0 * np.pi * 20000.0 / 44100.0).view(np.int32)
    dense = np.arange(int(lo), int(hi) + 1, 151, dtype=np.int32).view(f32)
    om = np.concatenate([om, dense])
    assert om.size >= 1_000_000 and om.min() > 6e-4 and om.max() < np.pi
    o64 = om.astype(np.float64)
    for name, fn in (("sinf", np.sin), ("cosf", np.cos)):
        got, want = _apply(shim, "v_" + name, om), fn(o64).astype(f32)
        d = _ulps(got, want)
        assert d.max() <= 1, (name, int(d.max()), float(om[int(np.argmax(d))]))
        print(f"{name}: {om.size} arguments, {float((d == 0).mean()) * 100:.2f} % correctly rounded, the rest one ulp off")
    # 10^(dB / 40): the CLAP range +-16 dB densely, and out to +-40 dB
    db = np.concatenate([rng.uniform(-16.0, 16.0, 700_000), rng.uniform(-40.0, 40.0, 300_000),
                         np.arange(-16000, 16001) * 1e-3]).astype(f32)
    y = db / f32(40.0)
    got, want = _apply(shim, "v_pow10", y), np.power(10.0, y.astype(np.float64)).astype(f32)
    d = _ulps(got, want)
    assert y.size >= 1_000_000 and d.max() <= 1, (int(d.max()), float(db[int(np.argmax(d))]))
    print(f"powf(10, y): {y.size} arguments, {float((d == 0).mean()) * 100:.2f} % correctly rounded, the rest one ulp off")


# ------------------------------------------------------------------------------------------------------------------
def _from_params_f32(t, fs, fc, q, gain_db, sn, cs, a):
    """biquad 0.4.2 Coefficients::<f32>::from_params, one IEEE f32 operation per numpy operation, left to right.
    sn, cs = sin / cos of omega, a = 10^(gain_db / 40): supplied by the caller."""
    two, one, half = f32(2.0), f32(1.0), f32(0.5)
    alpha = sn / (two * q)
    if t == 3:      # LowPass
        b0 = (one - cs) * half; b1 = one - cs; b2 = (one - cs) * half
        a0 = one + alpha; a1 = f32(-2.0) * cs; a2 = one - alpha
    elif t == 4:    # HighPass
        b0 = (one + cs) * half; b1 = -(one + cs); b2 = (one + cs) * half
        a0 = one + alpha; a1 = f32(-2.0) * cs; a2 = one - alpha
    elif t == 5:    # BandPass
        b0 = sn / two; b1 = np.zeros_like(sn); b2 = -(sn / two)
        a0 = one + alpha; a1 = f32(-2.0) * cs; a2 = one - alpha
    elif t == 6:    # Notch
        b0 = np.ones_like(sn); b1 = f32(-2.0) * cs; b2 = np.ones_like(sn)
        a0 = one + alpha; a1 = f32(-2.0) * cs; a2 = one - alpha
    elif t == 7:    # AllPass
        b0 = one - alpha; b1 = f32(-2.0) * cs; b2 = one + alpha
        a0 = one + alpha; a1 = f32(-2.0) * cs; a2 = one - alpha
    elif t == 1:    # LowShelf
        sa = np.sqrt(a)
        b0 = a * (((a + one) - (a - one) * cs) + (two * alpha) * sa)
        b1 = (two * a) * ((a - one) - (a + one) * cs)
        b2 = a * (((a + one) - (a - one) * cs) - (two * alpha) * sa)
        a0 = ((a + one) + (a - one) * cs) + (two * alpha) * sa
        a1 = f32(-2.0) * ((a - one) + (a + one) * cs)
        a2 = ((a + one) + (a - one) * cs) - (two * alpha) * sa
    elif t == 2:    # HighShelf
        sa = np.sqrt(a)
        b0 = a * (((a + one) + (a - one) * cs) + (two * alpha) * sa)
        b1 = (f32(-2.0) * a) * ((a - one) + (a + one) * cs)
        b2 = a * (((a + one) + (a - one) * cs) - (two * alpha) * sa)
        a0 = ((a + one) - (a - one) * cs) + (two * alpha) * sa
        a1 = two * ((a - one) - (a + one) * cs)
        a2 = ((a + one) - (a - one) * cs) - (two * alpha) * sa
    else:           # Peak
        b0 = one + alpha * a; b1 = f32(-2.0) * cs; b2 = one - alpha * a
        a0 = one + alpha / a; a1 = f32(-2.0) * cs; a2 = one - alpha / a
    out = np.stack([b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0], axis=-1)
    assert out.dtype == np.float32
    return out


def _sweep(n, seed):
    rng = np.random.default_rng(seed)
    fc = np.exp(rng.uniform(np.log(20.0), np.log(20000.0), n)).astype(f32)           # src/lib.rs:67-74
    q = rng.uniform(0.1, 10.0, n).astype(f32)                                         # :76-81
    g = rng.uniform(-16.0, 16.0, n).astype(f32)                                       # :83-86
    # the corners too
    fc[:4], q[:4], g[:4] = [20.0, 20000.0, 20.0, 20000.0], [0.1, 10.0, 10.0, 0.1], [-16.0, 16.0, 16.0, -16.0]
    fs = np.array([RATES[i] for i in rng.integers(0, len(RATES), n)], f32)
    return fs, fc, q, g


def _library_coeffs(t, fs, fc, q, g):
    from open_headstage_amd import _ffi
    L = _ffi.lib()
    out = np.empty((fc.size, 5), np.float32)
    buf = (C.c_float * 5)()
    for i in range(fc.size):
        rc = L.ohs_biquad_coeffs(int(t), C.c_float(float(fs[i])), C.c_float(float(fc[i])), C.c_float(float(q[i])),
                                 C.c_float(float(g[i])), buf)
        assert rc == 0, (t, fs[i], fc[i], q[i], g[i])
        out[i] = buf[:]
    return out


@pytest.mark.parametrize("t", range(8))
def test_formulas_are_biquad_0_4_2s_operation_for_operation(shim, t):
    """the library's routine == the numpy float32 re-implementation, BIT FOR BIT, given the same sin / cos / pow"""
    fs, fc, q, g = _sweep(3000, 100 + t)
    om = _omega_f32(fc, fs)
    sn, cs = _apply(shim, "v_sinf", om), _apply(shim, "v_cosf", om)
    a = _apply(shim, "v_pow10", g / f32(40.0))
    want = _from_params_f32(t, fs, fc, q, g, sn, cs, a)
    got = _library_coeffs(t, fs, fc, q, g)
    bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0]
    assert bad.size == 0, (t, bad[:5], got[bad[:1]], want[bad[:1]])


def _rbj_f64(t, om, q, gain_over_40, dcs=0.0, dsn=0.0, da=0.0):
    """Audio-EQ-Cookbook closed forms in f64; dcs / dsn / da: relative perturbations of cos w, sin w, A"""
    sn, cs = np.sin(om) * (1.0 + dsn), np.cos(om) * (1.0 + dcs)
    alpha = sn / (2.0 * q)
    A = np.power(10.0, gain_over_40) * (1.0 + da)
    if t == 3: b = [(1 - cs) / 2, 1 - cs, (1 - cs) / 2]; a = [1 + alpha, -2 * cs, 1 - alpha]
    elif t == 4: b = [(1 + cs) / 2, -(1 + cs), (1 + cs) / 2]; a = [1 + alpha, -2 * cs, 1 - alpha]
    elif t == 5: b = [sn / 2, 0 * sn, -sn / 2]; a = [1 + alpha, -2 * cs, 1 - alpha]
    elif t == 6: b = [1 + 0 * sn, -2 * cs, 1 + 0 * sn]; a = [1 + alpha, -2 * cs, 1 - alpha]
    elif t == 7: b = [1 - alpha, -2 * cs, 1 + alpha]; a = [1 + alpha, -2 * cs, 1 - alpha]
    elif t == 1:
        s2 = 2 * np.sqrt(A) * alpha
        b = [A * ((A + 1) - (A - 1) * cs + s2), 2 * A * ((A - 1) - (A + 1) * cs), A * ((A + 1) - (A - 1) * cs - s2)]
        a = [(A + 1) + (A - 1) * cs + s2, -2 * ((A - 1) + (A + 1) * cs), (A + 1) + (A - 1) * cs - s2]
    elif t == 2:
        s2 = 2 * np.sqrt(A) * alpha
        b = [A * ((A + 1) + (A - 1) * cs + s2), -2 * A * ((A - 1) + (A + 1) * cs), A * ((A + 1) + (A - 1) * cs - s2)]
        a = [(A + 1) - (A - 1) * cs + s2, 2 * ((A - 1) - (A + 1) * cs), (A + 1) - (A - 1) * cs - s2]
    else:
        b = [1 + alpha * A, -2 * cs, 1 - alpha * A]; a = [1 + alpha / A, -2 * cs, 1 - alpha / A]
    return np.stack([b[0] / a[0], b[1] / a[0], b[2] / a[0], a[1] / a[0], a[2] / a[0]], axis=-1)


def test_against_the_cookbook_in_f64_over_the_clap_ranges():
    n = 1500
    within2, total, worst_excess = 0, 0, 0.0
    eps = 2.0 ** -23                                    # one ulp of freedom in cos w, sin w and A (item 1 allows it)
    for t in range(8):
        fs, fc, q, g = _sweep(n, 500 + t)
        got = _library_coeffs(t, fs, fc, q, g).astype(np.float64)
        om = _omega_f32(fc, fs).astype(np.float64)      # (the f32 omega IS the crate's argument; its rounding is not at issue)
        q64, y64 = q.astype(np.float64), (g / f32(40.0)).astype(np.float64)
        ref = _rbj_f64(t, om, q64, y64)
        ulp = np.spacing(np.maximum(np.abs(ref), 1e-30).astype(f32)).astype(np.float64)
        err = np.abs(got - ref) / ulp
        # what one ulp in each of cos w, sin w, A moves the exact result by (the conditioning of the formula)
        spread = np.zeros_like(ref)
        for dcs in (-eps, 0.0, eps):
            for dsn in (-eps, 0.0, eps):
                for da in (-eps, 0.0, eps):
                    spread = np.maximum(spread, np.abs(_rbj_f64(t, om, q64, y64, dcs, dsn, da) - ref))
        allowed = spread / ulp * 1.5 + 4.0              # + the handful of roundings of the f32 evaluation itself
        excess = err - allowed
        assert excess.max() <= 0.0, (t, float(excess.max()), int(np.argmax(excess)) // 5)
        worst_excess = max(worst_excess, float((err / allowed).max()))
        within2 += int((err <= 2.0).sum()); total += err.size
        well = spread / ulp <= 1.0                      # well conditioned: one ulp in, at most one ulp out
        # (the shelves evaluate eight f32 operations per coefficient, the others three or four)
        assert err[well].max() <= (4.0 if t in (1, 2) else 2.5), (t, float(err[well].max()))
    print(f"{total} coefficients over the CLAP ranges: {100.0 * within2 / total:.1f} % within 2 ulp of the f64 cookbook; "
          f"the rest within their conditioning bound (worst use of it {worst_excess:.2f})")
    assert within2 / total >= 0.90
