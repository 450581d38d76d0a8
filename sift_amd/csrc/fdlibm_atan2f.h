// atan2f exactly as glibc <= 2.40 computes it (sysdeps/ieee754/flt-32/e_atan2f.c + s_atanf.c, the
// Sun fdlibm single-precision algorithm): the reference's alg::gradientOrientation
// (/root/reference/algorithms.cpp:113-116) calls libm's atan2f, and the device's own OCML atan2f
// rounds differently in the last bit.  Restating the published fdlibm algorithm with one IEEE
// operation per operator (compile with -ffp-contract=off) makes the device result bit-identical
// to the host libm's; tests/test_host_math.py checks this file against glibc on the CPU.
// The algorithm and its constants restate e_atan2f.c / s_atanf.c of Sun's fdlibm, whose licence
// asks for this notice to be preserved:
//
// ====================================================
// Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//
// Developed at SunPro, a Sun Microsystems, Inc. business.
// Permission to use, copy, modify, and distribute this
// software is freely granted, provided that this notice
// is preserved.
// ====================================================
//
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SIFT_HD __host__ __device__ inline
#else
#define SIFT_HD inline
#endif

namespace sift_hip {

SIFT_HD int32_t f2i(float x) { return __builtin_bit_cast(int32_t, x); }
SIFT_HD float i2f(int32_t i) { return __builtin_bit_cast(float, i); }
SIFT_HD float fabs_bits(float x) { return i2f(f2i(x) & 0x7fffffff); }

SIFT_HD float fdlibm_atanf(float x) {
    static constexpr float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    static constexpr float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
                aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
                aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
                aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const int32_t hx = f2i(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) { /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x; /* NaN */
        if (hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                 /* |x| < 0.4375 */
        if (ix < 0x31000000) return x;     /* |x| < 2^-29 */
        id = -1;
    } else {
        x = fabs_bits(x);
        if (ix < 0x3f980000) {             /* |x| < 1.1875 */
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else                 { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else                 { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float hi = id == 0 ? atanhi[0] : id == 1 ? atanhi[1] : id == 2 ? atanhi[2] : atanhi[3];
    const float lo = id == 0 ? atanlo[0] : id == 1 ? atanlo[1] : id == 2 ? atanlo[2] : atanlo[3];
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return (hx < 0) ? -r : r;
}

SIFT_HD float fdlibm_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
                pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = f2i(x), hy = f2i(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y; /* NaN */
    if (hx == 0x3f800000) return fdlibm_atanf(y);           /* x = 1.0 */
    const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);  /* 2*sign(x) + sign(y) */
    if (iy == 0) {
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = fdlibm_atanf(fabs_bits(y / x));
    switch (m) {
        case 0: return z;
        case 1: return i2f(f2i(z) ^ (int32_t)0x80000000);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// a / b, correctly rounded, for operands and quotients well inside the exponent range (round 6).  The division the compiler
// emits for `a / b` is this very sequence - reciprocal estimate, one Newton step on it, quotient, two residual corrections -
// wrapped in v_div_scale_f32 (twice) and v_div_fixup_f32, which only move the exponents out of harm's way and patch
// infinities, zeros and NaNs: 12 instructions where 9 do the arithmetic.  With |a|, |b| in [2^-60, 2^60] (the caller checks;
// everything else takes fdlibm_atan2f's own division) no intermediate leaves the normal range, the scaling is the identity
// and the result is the same correctly rounded quotient.  a == 0 gives the zero `a / b` gives.  On the host the estimate is
// 1.0f / b: any estimate within an ulp refines to the same quotient (tests/test_host_math.py compares with `/` on 10^8 pairs).
SIFT_HD float div_in_range(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(b);
#else
    float r = 1.0f / b;
#endif
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    float rem = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(rem, r, q);
}

// Branch-free evaluation of the same algorithm for the common case, for the GPU: the four argument
// reductions of atanf differ only in the numerator and denominator of their single division, so they are
// selected first and divided once; the polynomial and the quadrant fix-up are the same operations in the
// same order as above.  Everything else (NaN, infinities, zeros, x == 1, |y/x| outside [2^-29, 2^25),
// exponent gap > 60) is left to fdlibm_atan2f.
SIFT_HD bool fdlibm_atan2f_common(float y, float x, float& out) {
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = f2i(x), hy = f2i(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    // (round 6: both divisions as div_in_range - |x|, |y| in [2^-60, 2^60], which also keeps the exponent gap k within
    // +-120 > 60, checked below; the reduced argument's numerator and denominator lie in [2^-29, 2^25] by construction)
    const int32_t k = (iy - ix) >> 23;
    const float t = fabs_bits(div_in_range(y, x));
    const int32_t it = f2i(t);
    const bool common = (uint32_t)(ix - 0x21800000) <= (uint32_t)(0x5d800000 - 0x21800000) &&
                        (uint32_t)(iy - 0x21800000) <= (uint32_t)(0x5d800000 - 0x21800000) && hx != 0x3f800000 && k <= 60 &&
                        k >= -60 && it < 0x4c000000 && it >= 0x31000000;
    // argument reduction: x' = num / den
    const bool c_small = it < 0x3ee00000, c0 = it < 0x3f300000, c1 = it < 0x3f980000, c2 = it < 0x401c0000;
    const float num = c_small ? t : c0 ? 2.0f * t - 1.0f : c1 ? t - 1.0f : c2 ? t - 1.5f : -1.0f;
    const float den = c_small ? 1.0f : c0 ? 2.0f + t : c1 ? t + 1.0f : c2 ? 1.0f + 1.5f * t : t;
    const float hi = c0 ? 4.6364760399e-01f : c1 ? 7.8539812565e-01f : c2 ? 9.8279368877e-01f : 1.5707962513e+00f;
    const float lo = c0 ? 5.0121582440e-09f : c1 ? 3.7748947079e-08f : c2 ? 3.4473217170e-08f : 7.5497894159e-08f;
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
                aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
                aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
                aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const float xr = div_in_range(num, den);
    const float z = xr * xr;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float p = xr * (s1 + s2);
    const float at = c_small ? xr - p : hi - ((p - lo) - xr);   // atanf(|y/x|) >= 0
    const float zz = at - pi_lo;
    const float r = hx >= 0 ? at : pi - zz;                       // m = 0 / 2 (y >= 0)
    const float rn = hx >= 0 ? i2f(f2i(at) ^ (int32_t)0x80000000) : zz - pi;   // m = 1 / 3 (y < 0)
    out = hy >= 0 ? r : rn;
    return common;
}

SIFT_HD float fdlibm_atan2f_sel(float y, float x) {
    float r;
    if (fdlibm_atan2f_common(y, x, r)) return r;
    return fdlibm_atan2f(y, x);
}

}  // namespace sift_hip
