// alg::gradientMagnitude (/root/reference/algorithms.cpp:108-111) for the gradient kernels, shared with the CPU build the
// `-m "not gpu"` tests check (hostmath_capi.cpp):  (float)sqrt((double)dx * dx + (double)dy * dy)  - std::pow(float, int) and
// std::sqrt of the sum run in double, the result is rounded to float on return.
#pragma once
#include <stdint.h>

#include "fdlibm_atan2f.h"   // SIFT_HD

namespace sift_hip {

// Round 6.  The compiler's correctly rounded double square root is 22 instructions of which 17 run at the double-precision rate:
// exponent scaling for tiny and huge arguments (the square of a float is neither, in double), a reciprocal-square-root estimate
// and TWO Newton / Goldschmidt rounds - needed to round correctly IN DOUBLE.  The reference only keeps 24 bits of the result.
// So: the estimate (v_rsq_f64: 2^29 ulp, i.e. relative error e <= 2^-24), ONE round - g = S y, h = y / 2, g1 = g + (S - g g) h =
// sqrt(S) (1 - e^2) up to the roundings of three double operations: |g1 - sqrt(S)| < 2^-46 sqrt(S) - and (float)g1 IS the
// reference's (float)sqrt(S) unless a float rounding boundary (the midpoint of two neighbouring floats: the 29 mantissa bits a
// float drops = 0x10000000) lies within that distance of g1, which the dropped bits of g1 show.  Those inputs - 2^-20 of all:
// ~60 pixels of a batch of 32 frames - and S == 0, infinities and NaNs take the correctly rounded square root.  Exact for every
// input by construction; tests/test_host_math.py runs the host build (estimate truncated to 24 bits) against numpy's correctly
// rounded sqrt on 10^8 pairs and on pairs built to land within 2^-48 of a midpoint, the GPU tests the kernel on the same.
SIFT_HD float gradient_magnitude(float dx, float dy) {
    const double S = (double)dx * (double)dx + (double)dy * (double)dy;
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(S);
#else
    const double y = (double)(float)(1.0 / __builtin_sqrt(S));   // 24 bits, like the device's estimate at its worst
#endif
    const double g = S * y, h = 0.5 * y;
    const double r = __builtin_fma(-g, g, S);
    const double g1 = __builtin_fma(r, h, g);
    const uint32_t dropped = (uint32_t)(uint64_t)__builtin_bit_cast(int64_t, g1) & 0x1fffffffu;
    // within 2^9 units of the last double bit (2^-43 g1) of a midpoint, or not a positive finite number: the exact routine
    const bool sure = (uint32_t)(dropped - (0x10000000u - 512u)) > 1024u && S > 0.0 && g1 < 1.0e300;
    float m = (float)g1;
#if defined(__HIP_DEVICE_COMPILE__)
    // a wave-uniform branch (the compiler would otherwise evaluate both square roots for every pixel and select)
    if (__builtin_amdgcn_ballot_w64(!sure) != 0ull) {
        double sb = S;
        asm volatile("" : "+v"(sb));   // ... and would hoist the exact routine out of the branch without this
        if (!sure) m = (float)__builtin_sqrt(sb);
    }
#else
    if (!sure) m = (float)__builtin_sqrt(S);
#endif
    return m;
}

}  // namespace sift_hip
