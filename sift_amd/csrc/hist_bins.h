// Histogram bin index of alg::orientationHistogram8 (/root/reference/algorithms.cpp:143-145):
//     u16_t i = std::floor(orientation / 45);  i = i % 7;
// shared by the descriptor kernels and the CPU test build (sift_amd/csrc/hostmath_capi.cpp).
#pragma once
#include <stdint.h>

#include "fdlibm_atan2f.h"   // SIFT_HD

namespace sift_hip {

// float -> u16_t as x86 compiles it: cvttss2si (0x80000000 out of range / NaN), low 16 bits
SIFT_HD unsigned f32_to_u16_x86(float v) {
    int i;
    if (v > -2147483904.0f && v < 2147483648.0f)
        i = (int)v;
    else
        i = (int)0x80000000;
    return (unsigned)i & 0xffffu;
}

// the reference's own arithmetic: IEEE division
SIFT_HD unsigned hist8_bin_div(float v) { return f32_to_u16_x86(__builtin_floorf(v / 45.0f)) % 7u; }

// The quotient without the division sequence: q0 = v * RN(1/45), one Newton correction with the exact residual
// (Markstein).  For every one of the 2^32 float inputs the bin equals hist8_bin_div's (tests/test_host_math.py runs all of
// them on the host); the quotient itself differs only for -0.0 and the infinities, which land in the same bin.  Needs fused
// multiply-adds and f32 denormals, both on in this build (the explicit fma is not subject to -ffp-contract=off).
SIFT_HD unsigned hist8_bin(float v) {
    const float r = 1.0f / 45.0f;
    const float q0 = v * r;
    const float e = __builtin_fmaf(-q0, 45.0f, v);
    const float q = __builtin_fmaf(e, r, q0);
    return f32_to_u16_x86(__builtin_floorf(q)) % 7u;
}

}  // namespace sift_hip
