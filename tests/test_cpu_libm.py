"""The coefficient code's sinf / cosf / powf (open_headstage_amd/csrc/libm_musl.h, a restatement of what the `libm`
crate 0.1.4 -- a port of musl -- computes for biquad 0.4.2; Cargo.lock:314-320,1521-1522).  PARITY UNPINNED: the
crate is not on this machine.  What can be checked here: the product's and the oracle's copies are the same text,
they agree with the platform's libm to one ulp over everything the coefficient code can ask for (omega in (0, pi],
10^(dB/40) for |dB| <= 40), they are exact on the values the algorithms guarantee, and ohs_biquad_coeffs uses them
(its result does not follow the platform's libm where the two differ)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = os.path.join(ROOT, "open_headstage_amd", "csrc", "libm_musl.h")
ORACLE = os.path.join(ROOT, "oracle", "libm_musl.h")

SHIM = r'''
#include "%s"
void v_sinf(const float *x, float *y, long n) { for (long i = 0; i < n; ++i) y[i] = ohs_musl_sinf(x[i]); }
void v_cosf(const float *x, float *y, long n) { for (long i = 0; i < n; ++i) y[i] = ohs_musl_cosf(x[i]); }
void v_pow10(const float *x, float *y, long n) { for (long i = 0; i < n; ++i) y[i] = ohs_musl_powf(10.0f, x[i]); }
void g_sinf(const float *x, float *y, long n) { for (long i = 0; i < n; ++i) y[i] = sinf(x[i]); }
void g_cosf(const float *x, float *y, long n) { for (long i = 0; i < n; ++i) y[i] = cosf(x[i]); }
void g_pow10(const float *x, float *y, long n) { for (long i = 0; i < n; ++i) y[i] = powf(10.0f, x[i]); }
'''


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    d = tmp_path_factory.mktemp("libm")
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
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def test_product_and_oracle_hold_the_same_text():
    assert open(PRODUCT).read() == open(ORACLE).read()


def test_restatement_within_one_ulp_of_the_platform_libm(shim):
    # every float in (2^-20, pi] in steps of 97 ulps; dB / 40 for |dB| <= 40 in steps of 1e-4
    lo, hi = np.float32(2.0 ** -20).view(np.int32), np.float32(np.pi).view(np.int32)
    om = np.arange(int(lo), int(hi) + 1, 97, dtype=np.int32).view(np.float32)
    for f in ("sinf", "cosf"):
        d = _ulps(_apply(shim, "v_" + f, om), _apply(shim, "g_" + f, om))
        assert d.max() <= 1, (f, int(d.max()))
    y = (np.arange(-400000, 400001, dtype=np.float64) * 1e-4 / 40.0).astype(np.float32)
    d = _ulps(_apply(shim, "v_pow10", y), _apply(shim, "g_pow10", y))
    assert d.max() <= 1, int(d.max())
    # the restatement is not the platform's routine: some results differ (else this test pins nothing)
    assert (d != 0).any()


def test_exact_values(shim):
    one = np.array([0.0, -0.0], np.float32)
    assert np.array_equal(_apply(shim, "v_pow10", one), np.array([1.0, 1.0], np.float32))          # x**0 = 1
    assert np.array_equal(_apply(shim, "v_pow10", np.array([1.0, -1.0, 2.0], np.float32)),
                          np.array([10.0, np.float32(1.0) / np.float32(10.0), 100.0], np.float32))   # y = +-1, 2
    tiny = np.array([1e-5, 2.0 ** -13], np.float32)
    assert np.array_equal(_apply(shim, "v_sinf", tiny), tiny)                                       # |x| < 2^-12
    assert np.array_equal(_apply(shim, "v_cosf", tiny), np.ones(2, np.float32))
    # the four reduction ranges meet without a jump
    edges = np.array([0x3f490fda, 0x3f490fdb, 0x4016cbe3, 0x4016cbe4, 0x407b53d1, 0x407b53d2, 0x40afeddf, 0x40afede0,
                      0x40e231d5], np.uint32).view(np.float32)
    for f in ("sinf", "cosf"):
        assert _ulps(_apply(shim, "v_" + f, edges), _apply(shim, "g_" + f, edges)).max() <= 1


def test_biquad_coeffs_use_the_restatement_not_the_platform_libm(shim):
    """Pick a shelf gain where this host's powf and the restatement differ by an ulp: the library's coefficients must
    be the ones the restatement gives (recomputed here in float32 with numpy, biquad 0.4.2's formulas)."""
    from open_headstage_amd import FilterType, biquad_coefficients
    y = (np.arange(-16000, 16001, dtype=np.float64) * 1e-3 / 40.0).astype(np.float32)
    a_r, a_g = _apply(shim, "v_pow10", y), _apply(shim, "g_pow10", y)
    idx = np.nonzero(a_r != a_g)[0]
    assert idx.size, "no differing gain found"
    f32 = np.float32
    checked = 0
    for i in idx[:50]:
        gain_db = f32(y[i]) * f32(40.0)
        if f32(gain_db / f32(40.0)) != y[i]:
            continue                        # the division in the formula must reproduce y exactly
        fs, fc, q = f32(48000.0), f32(1000.0), f32(1.41)
        omega = f32(2.0) * f32(3.14159265358979323846) * fc / fs
        sn, cs = _apply(shim, "v_sinf", [omega])[0], _apply(shim, "v_cosf", [omega])[0]
        alpha = sn / (f32(2.0) * q)
        got = biquad_coefficients(FilterType.Peak, 48000.0, 1000.0, 1.41, float(gain_db))

        def peak(a):
            b0, b1, b2 = f32(1.0) + alpha * a, f32(-2.0) * cs, f32(1.0) - alpha * a
            a0, a1, a2 = f32(1.0) + alpha / a, f32(-2.0) * cs, f32(1.0) - alpha / a
            return np.array([b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0], np.float32)

        assert np.array_equal(got, peak(a_r[i])), (float(gain_db), got, peak(a_r[i]))
        checked += 1
    assert checked >= 3
