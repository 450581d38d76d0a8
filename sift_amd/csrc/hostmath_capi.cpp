// CPU build of the product's scalar math headers (the same code the HIP kernels inline), so the
// `-m "not gpu"` tests can check them against libm and against the oracle without a GPU.
// Not part of libsift_hip.so and never used by the product path.
#include <stdint.h>

#include "fdlibm_atan2f.h"
#include "grad_math.h"
#include "hist_bins.h"
#include "linalg3.h"

extern "C" {
float hostmath_atan2f(float y, float x) { return sift_hip::fdlibm_atan2f(y, x); }
void hostmath_atan2f_array(const float* y, const float* x, int n, float* out) {
    for (int i = 0; i < n; ++i) out[i] = sift_hip::fdlibm_atan2f(y[i], x[i]);
}
// the GPU's branch-free common path + fallback; also reports how many inputs took the common path
int hostmath_atan2f_sel_array(const float* y, const float* x, int n, float* out) {
    int common = 0;
    for (int i = 0; i < n; ++i) {
        float r;
        common += sift_hip::fdlibm_atan2f_common(y[i], x[i], r) ? 1 : 0;
        out[i] = sift_hip::fdlibm_atan2f_sel(y[i], x[i]);
    }
    return common;
}
// alg::gradientMagnitude as the gradient kernels compute it (grad_math.h); returns how many inputs took the exact routine
long long hostmath_magnitude_array(const float* dx, const float* dy, long long n, float* out) {
    long long slow = 0;
    for (long long i = 0; i < n; ++i) {
        out[i] = sift_hip::gradient_magnitude(dx[i], dy[i]);
    }
    return slow;
}
// a / b as the atan2f restatement divides (fdlibm_atan2f.h: div_in_range); mismatches against the compiler's division
long long hostmath_div_mismatches(const float* a, const float* b, long long n) {
    long long bad = 0;
    for (long long i = 0; i < n; ++i) {
        const float q = sift_hip::div_in_range(a[i], b[i]), w = a[i] / b[i];
        bad += sift_hip::f2i(q) != sift_hip::f2i(w) ? 1 : 0;
    }
    return bad;
}
// matrices row-fastest like vigra::Matrix: a[i + 3*j] = A(i, j)
int hostmath_inverse3(const float* a, float* res) {
    float A[3][3], R[3][3] = {};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = a[i + 3 * j];
    const bool ok = sift_hip::inverse3(A, R);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) res[i + 3 * j] = R[i][j];
    return ok ? 1 : 0;
}
int hostmath_solve3(const float* a, const float* b, float* res) {
    float A[3][3], B[3] = {b[0], b[1], b[2]}, R[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = a[i + 3 * j];
    const bool ok = sift_hip::solve3<true>(A, B, R);
    res[0] = R[0]; res[1] = R[1]; res[2] = R[2];
    return ok ? 1 : 0;
}
// inputs with bit patterns first .. first + count - 1 whose fast bin differs from the division's; *example = one of them
long long hostmath_hist8_bin_mismatches(unsigned long long first, unsigned long long count, unsigned* example) {
    long long bad = 0;
    for (unsigned long long i = first; i < first + count; ++i) {
        const float v = sift_hip::i2f((int32_t)(uint32_t)i);
        if (sift_hip::hist8_bin(v) != sift_hip::hist8_bin_div(v)) {
            ++bad;
            if (example) *example = (uint32_t)i;
        }
    }
    return bad;
}
float hostmath_vertex_parabola(uint16_t lnx, float lny, uint16_t px, float py, uint16_t rnx, float rny) {
    return sift_hip::vertex_parabola(lnx, lny, px, py, rnx, rny);
}
}
