/*
 * libm_musl.h -- sinf / cosf / powf as the reference's coefficient code evaluates them.
 *
 * BiquadFilter::update_coeffs (src/dsp/parametric_eq.rs:105-111) calls biquad 0.4.2's
 * Coefficients::<f32>::from_params, which takes its f32 trigonometry and powf from the `libm` crate 0.1.4
 * (Cargo.lock:314-320,1521-1522) -- a Rust port of musl's (FreeBSD msun's) routines, NOT the platform libm.
 * The EQ's output is within 1e-6 of the reference only with bit-identical coefficients (one ulp in one coefficient
 * of the 10-band table moves the output by ~2e-4 relative RMS, tests/test_gpu_parity.py), so the convenience path
 * ohs_biquad_coeffs / ohs_eq_update_band must not depend on which glibc the host happens to run.  These are
 * restatements of the published algorithms:
 *     sinf, cosf  msun s_sinf.c / s_cosf.c with k_sinf.c / k_cosf.c: argument reduction by exact multiples of
 *                 pi/2 in double, degree-9 / degree-8 minimax polynomials in double, one rounding to float
 *     powf        msun e_powf.c (the fdlibm algorithm: log2(x) to ~32 bits as t1 + t2, y * log2(x) split in two,
 *                 2^z by a degree-5 Remez rational)
 * PARITY UNPINNED: neither the crate nor musl is on this machine; the constants and operation order below are from
 * the published sources as the author knows them, checked here against glibc (<= 1 ulp, tests/test_cpu_libm.py) and
 * against correctly rounded results over > 10^6 arguments of the coefficient domain (<= 1 ulp,
 * tests/test_cpu_coeff_pins.py).
 * Arguments outside what the coefficient code produces (|x| > 9 pi / 4 for sinf / cosf; x <= 0, non-finite or
 * subnormal cases of powf) fall through to the platform's libm.
 *
 * Plain C99, no dependencies: the same text builds into the library (hipcc, host side) and is held, as a separate
 * copy, by the CPU oracle (oracle/libm_musl.h) -- test infrastructure shares no file with the product.
 * Build WITHOUT fused contraction (-ffp-contract=off or the pragmas below): every operation rounds by itself.
 *
 * UPSTREAM NOTICES.  The constants, the argument-reduction scheme and the operation order restated below are those of
 * FreeBSD msun (s_sinf.c, s_cosf.c, k_sinf.c, k_cosf.c, e_powf.c) as carried by musl libc and by the Rust `libm` crate;
 * this file is a derived work of them and carries their notices:
 *
 *   ====================================================
 *   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *
 *   Developed at SunPro, a Sun Microsystems, Inc. business.
 *   Permission to use, copy, modify, and distribute this
 *   software is freely granted, provided that this notice
 *   is preserved.
 *   ====================================================
 *   (msun: "Conversion to float by Ian Lance Taylor, Cygnus Support, ian@cygnus.com.  Optimized by Bruce D. Evans.")
 *
 *   musl libc -- Copyright (c) 2005-2020 Rich Felker, et al.  Licensed under the MIT license:
 *   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
 *   documentation files (the "Software"), to deal in the Software without restriction, including without limitation
 *   the rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and
 *   to permit persons to whom the Software is furnished to do so, subject to the following conditions: The above
 *   copyright notice and this permission notice shall be included in all copies or substantial portions of the
 *   Software.  THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT
 *   LIMITED TO THE WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT.  IN NO EVENT
 *   SHALL THE AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF
 *   CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS
 *   IN THE SOFTWARE.
 *
 *   `libm` crate 0.1.4 (rust-lang/libm) -- dual-licensed MIT / Apache-2.0, Copyright (c) 2018 Jorge Aparicio; its
 *   sinf / cosf / powf are ports of the musl files above and keep the Sun notice.
 */
#ifndef OHS_LIBM_MUSL_H
#define OHS_LIBM_MUSL_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
#endif

static inline uint32_t ohs_musl_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float ohs_musl_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* k_sinf.c: |x| ~<= pi/4, x in double, result rounded once */
static inline float ohs_musl_sindf(double x)
{
    static const double S1 = -0x15555554cbac77.0p-55, /* -0.166666666416265235595 */
                        S2 = 0x111110896efbb2.0p-59,  /*  0.0083333293858894631756 */
                        S3 = -0x1a00f9e2cae774.0p-65, /* -0.000198393348360966317347 */
                        S4 = 0x16cd878c3b46a7.0p-71;  /*  0.0000027183114939898219064 */
    const double z = x * x;
    const double w = z * z;
    const double r = S3 + z * S4;
    const double s = z * x;
    return (float)((x + s * (S1 + z * S2)) + s * w * r);
}

/* k_cosf.c */
static inline float ohs_musl_cosdf(double x)
{
    static const double C0 = -0x1ffffffd0c5e81.0p-54, /* -0.499999997251031003120 */
                        C1 = 0x155553e1053a42.0p-57,  /*  0.0416666233237390631894 */
                        C2 = -0x16c087e80f1e27.0p-62, /* -0.00138867637746099294692 */
                        C3 = 0x199342e0ee5069.0p-68;  /*  0.0000243904487962774090654 */
    const double z = x * x;
    const double w = z * z;
    const double r = C2 + z * C3;
    return (float)(((1.0 + z * C0) + w * C1) + (w * z) * r);
}

#define OHS_MUSL_PIO2 1.57079632679489661923 /* M_PI_2 as a double: 0x3FF921FB54442D18 */

static inline float ohs_musl_sinf(float x)
{
    const double s1pio2 = 1 * OHS_MUSL_PIO2, s2pio2 = 2 * OHS_MUSL_PIO2, s3pio2 = 3 * OHS_MUSL_PIO2, s4pio2 = 4 * OHS_MUSL_PIO2;
    uint32_t ix = ohs_musl_f2u(x);
    const int sign = (int)(ix >> 31);
    ix &= 0x7fffffff;
    if (ix <= 0x3f490fda) {                 /* |x| ~<= pi/4 */
        if (ix < 0x39800000) return x;      /* |x| < 2**-12 */
        return ohs_musl_sindf(x);
    }
    if (ix <= 0x407b53d1) {                 /* |x| ~<= 5*pi/4 */
        if (ix <= 0x4016cbe3) {             /* |x| ~<= 3pi/4 */
            if (sign) return -ohs_musl_cosdf(x + s1pio2);
            return ohs_musl_cosdf(x - s1pio2);
        }
        return ohs_musl_sindf(sign ? -(x + s2pio2) : -(x - s2pio2));
    }
    if (ix <= 0x40e231d5) {                 /* |x| ~<= 9*pi/4 */
        if (ix <= 0x40afeddf) {             /* |x| ~<= 7*pi/4 */
            if (sign) return ohs_musl_cosdf(x + s3pio2);
            return -ohs_musl_cosdf(x - s3pio2);
        }
        return ohs_musl_sindf(sign ? x + s4pio2 : x - s4pio2);
    }
    return sinf(x);     /* general argument reduction: not reachable from the coefficient code (omega <= pi) */
}

static inline float ohs_musl_cosf(float x)
{
    const double c1pio2 = 1 * OHS_MUSL_PIO2, c2pio2 = 2 * OHS_MUSL_PIO2, c3pio2 = 3 * OHS_MUSL_PIO2, c4pio2 = 4 * OHS_MUSL_PIO2;
    uint32_t ix = ohs_musl_f2u(x);
    const int sign = (int)(ix >> 31);
    ix &= 0x7fffffff;
    if (ix <= 0x3f490fda) {                 /* |x| ~<= pi/4 */
        if (ix < 0x39800000) return 1.0f;   /* |x| < 2**-12 */
        return ohs_musl_cosdf(x);
    }
    if (ix <= 0x407b53d1) {                 /* |x| ~<= 5*pi/4 */
        if (ix > 0x4016cbe3)                /* |x|  ~> 3*pi/4 */
            return -ohs_musl_cosdf(sign ? x + c2pio2 : x - c2pio2);
        if (sign) return ohs_musl_sindf(x + c1pio2);
        return ohs_musl_sindf(c1pio2 - x);
    }
    if (ix <= 0x40e231d5) {                 /* |x| ~<= 9*pi/4 */
        if (ix > 0x40afeddf)                /* |x| ~> 7*pi/4 */
            return ohs_musl_cosdf(sign ? x + c4pio2 : x - c4pio2);
        if (sign) return ohs_musl_sindf(-x - c3pio2);
        return ohs_musl_sindf(x - c3pio2);
    }
    return cosf(x);
}

/* e_powf.c, the path positive finite normal x and finite y of moderate size take; everything else -> libm */
static inline float ohs_musl_powf(float x, float y)
{
    static const float bp[2] = {1.0f, 1.5f},
                       dp_h[2] = {0.0f, 5.84960938e-01f},   /* 0x3f15c000 */
                       dp_l[2] = {0.0f, 1.56322085e-06f};   /* 0x35d1cfdc */
    static const float L1 = 6.0000002384e-01f,  /* 0x3f19999a */
                       L2 = 4.2857143283e-01f,  /* 0x3edb6db7 */
                       L3 = 3.3333334327e-01f,  /* 0x3eaaaaab */
                       L4 = 2.7272811532e-01f,  /* 0x3e8ba305 */
                       L5 = 2.3066075146e-01f,  /* 0x3e6c3255 */
                       L6 = 2.0697501302e-01f,  /* 0x3e53f142 */
                       P1 = 1.6666667163e-01f,  /* 0x3e2aaaab */
                       P2 = -2.7777778450e-03f, /* 0xbb360b61 */
                       P3 = 6.6137559770e-05f,  /* 0x388ab355 */
                       P4 = -1.6533901999e-06f, /* 0xb5ddea0e */
                       P5 = 4.1381369442e-08f,  /* 0x3331bb4c */
                       lg2 = 6.9314718246e-01f,   /* 0x3f317218 */
                       lg2_h = 6.93145752e-01f,   /* 0x3f317200 */
                       lg2_l = 1.42860654e-06f,   /* 0x35bfbe8c */
                       ovt = 4.2995665694e-08f,   /* -(128-log2(ovfl+.5ulp)) */
                       cp = 9.6179670095e-01f,    /* 0x3f76384f =2/(3ln2) */
                       cp_h = 9.6191406250e-01f,  /* 0x3f764000 =12b cp */
                       cp_l = -1.1736857402e-04f; /* 0xb8f623c6 =tail of cp_h */
    const int32_t hx = (int32_t)ohs_musl_f2u(x), hy = (int32_t)ohs_musl_f2u(y);
    int32_t ix = hx & 0x7fffffff;
    const int32_t iy = hy & 0x7fffffff;
    if (iy == 0) return 1.0f;               /* x**0 = 1, even if x is NaN */
    if (hx == 0x3f800000) return 1.0f;      /* 1**y = 1, even if y is NaN */
    /* anything but a positive normal finite x and a finite |y| <= 2**27: the platform's powf */
    if (hx <= 0 || ix >= 0x7f800000 || ix < 0x00800000 || iy >= 0x7f800000 || iy > 0x4d000000) return powf(x, y);
    if (iy == 0x3f800000) return hy >= 0 ? x : 1.0f / x;   /* y is +-1 */
    if (hy == 0x40000000) return x * x;                     /* y is 2 */
    if (hy == 0x3f000000) return sqrtf(x);                  /* y is 0.5 (x >= 0 here) */

    float ax = x, z, z_h, z_l, p_h, p_l, y1, t1, t2, r, s, t, u, v, w;
    int32_t j, k, n, is;
    {
        float s2, s_h, s_l, t_h, t_l;
        n = (ix >> 23) - 0x7f;
        j = ix & 0x007fffff;
        /* determine interval */
        ix = j | 0x3f800000;            /* normalize ix */
        if (j <= 0x1cc471) k = 0;       /* |x|<sqrt(3/2) */
        else if (j < 0x5db3d7) k = 1;   /* |x|<sqrt(3)   */
        else { k = 0; n += 1; ix -= 0x00800000; }
        ax = ohs_musl_u2f((uint32_t)ix);

        /* compute s = s_h+s_l = (x-1)/(x+1) or (x-1.5)/(x+1.5) */
        u = ax - bp[k];
        v = 1.0f / (ax + bp[k]);
        s = u * v;
        s_h = ohs_musl_u2f(ohs_musl_f2u(s) & 0xfffff000);
        /* t_h=ax+bp[k] High */
        is = (int32_t)((((uint32_t)ix >> 1) & 0xfffff000) | 0x20000000);
        t_h = ohs_musl_u2f((uint32_t)(is + 0x00400000 + (k << 21)));
        t_l = ax - (t_h - bp[k]);
        s_l = v * ((u - s_h * t_h) - s_h * t_l);
        /* compute log(ax) */
        s2 = s * s;
        r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
        r += s_l * (s_h + s);
        s2 = s_h * s_h;
        t_h = 3.0f + s2 + r;
        t_h = ohs_musl_u2f(ohs_musl_f2u(t_h) & 0xfffff000);
        t_l = r - ((t_h - 3.0f) - s2);
        /* u+v = s*(1+...) */
        u = s_h * t_h;
        v = s_l * t_h + t_l * s;
        /* 2/(3log2)*(s+...) */
        p_h = u + v;
        p_h = ohs_musl_u2f(ohs_musl_f2u(p_h) & 0xfffff000);
        p_l = v - (p_h - u);
        z_h = cp_h * p_h;               /* cp_h+cp_l = 2/(3*log2) */
        z_l = cp_l * p_h + p_l * cp + dp_l[k];
        /* log2(ax) = (s+..)*2/(3*log2) = n + dp_h + z_h + z_l */
        t = (float)n;
        t1 = (((z_h + z_l) + dp_h[k]) + t);
        t1 = ohs_musl_u2f(ohs_musl_f2u(t1) & 0xfffff000);
        t2 = z_l - (((t1 - t) - dp_h[k]) - z_h);
    }
    /* split up y into y1+y2 and compute (y1+y2)*(t1+t2) */
    y1 = ohs_musl_u2f(ohs_musl_f2u(y) & 0xfffff000);
    p_l = (y - y1) * t1 + y * t2;
    p_h = y1 * t1;
    z = p_l + p_h;
    j = (int32_t)ohs_musl_f2u(z);
    if (j > 0x43000000) return powf(x, y);                              /* overflow */
    if (j == 0x43000000) { if (p_l + ovt > z - p_h) return powf(x, y); }
    else if ((j & 0x7fffffff) > 0x43160000) return powf(x, y);          /* underflow */
    else if ((uint32_t)j == 0xc3160000u) { if (p_l <= z - p_h) return powf(x, y); }
    /* compute 2**(p_h+p_l) */
    {
        const int32_t i = j & 0x7fffffff;
        k = (i >> 23) - 0x7f;
        n = 0;
        if (i > 0x3f000000) {           /* if |z| > 0.5, set n = [z+0.5] */
            n = j + (0x00800000 >> (k + 1));
            k = ((n & 0x7fffffff) >> 23) - 0x7f;    /* new k for n */
            t = ohs_musl_u2f((uint32_t)(n & ~(0x007fffff >> k)));
            n = ((n & 0x007fffff) | 0x00800000) >> (23 - k);
            if (j < 0) n = -n;
            p_h -= t;
        }
    }
    t = p_l + p_h;
    t = ohs_musl_u2f(ohs_musl_f2u(t) & 0xffff8000);
    u = t * lg2_h;
    v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
    z = u + v;
    w = v - (z - u);
    t = z * z;
    t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    r = (z * t1) / (t1 - 2.0f) - (w + z * w);
    z = 1.0f - (r - z);
    j = (int32_t)ohs_musl_f2u(z);
    j += (int32_t)((uint32_t)n << 23);
    if ((j >> 23) <= 0) return powf(x, y);      /* subnormal output */
    return ohs_musl_u2f((uint32_t)j);
}

#if !defined(__clang__) && defined(__GNUC__)
#pragma GCC pop_options
#endif

#endif /* OHS_LIBM_MUSL_H */
