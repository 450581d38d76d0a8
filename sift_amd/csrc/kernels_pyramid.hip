// Gaussian pyramid kernels: separable Gaussian blur (+ fused DoG), nearest-neighbour resampling.
//
// Replaces alg::convolveWithGauss (/root/reference/algorithms.cpp:10-22: Vigra
// separableConvolveX then separableConvolveY with BORDER_TREATMENT_REFLECT), alg::dog
// (algorithms.cpp:52-64) and the resizeImageNoInterpolation calls of alg::reduceToNextLevel /
// increaseToNextLevel (algorithms.cpp:24-49) for the loops of Sift::_createDOGs
// (/root/reference/sift.cpp:381-417).
//
// Bit-exactness contract: every output pixel is  sum_{p = x-r .. x+r, ascending} tap * src[reflect(p)]
// accumulated in float from 0.0f with one rounding per multiply and per add (no FMA: this file is
// compiled with -ffp-contract=off and repeats it in a pragma), the row pass result is rounded to
// float before the column pass consumes it (the reference stores it in `tmp`), and
// reflect(p) = -p for p < 0, 2(w-1)-p for p >= w.
#include "common.h"

#pragma clang fp contract(off)

namespace sift_hip {

__device__ __forceinline__ int reflect_clamp(int p, int n) {
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * (n - 1) - p : p;
    // lanes that only feed outputs outside the image may still be out of range: keep them legal
    return p < 0 ? 0 : (p >= n ? n - 1 : p);
}

// ---------------------------------------------------------------------------------------------
// Fused blur: one workgroup produces a 64 x TH output tile.
//   1. row-coalesced HBM loads of the (64+2R) x (TH+2R) source tile (halo reflected at the image
//      border) into LDS,
//   2. row pass LDS -> registers (each thread 4 consecutive outputs, ds_read_b128 windows) -> LDS,
//   3. column pass LDS -> registers (each thread TH/4 consecutive rows of one column, conflict-free
//      ds_read_b32), DoG = 128 + (blurred - source) from the source tile still in LDS,
//   4. coalesced stores.
// The intermediate never touches HBM: 4 B read + 4 B (8 B with DoG) written per pixel.
// Taps are read with wave-uniform constant indices => scalar loads into SGPRs.
// Workgroup ids are remapped so that each XCD (ids dealt round-robin over 8 XCDs) walks a
// contiguous run of tiles: neighbouring tiles' halos then hit that XCD's L2.
// ---------------------------------------------------------------------------------------------
template <int R, int TH, bool DOG>
__global__ __launch_bounds__(256) void blur_fused_kernel(const float* __restrict__ in,
                                                         float* __restrict__ out,
                                                         float* __restrict__ dog, int w, int h,
                                                         int tiles_x, int tiles_y,
                                                         const float* __restrict__ taps) {
    constexpr int TW = 64;
    constexpr int SW = TW + 2 * R;
    constexpr int SH = TH + 2 * R;
    constexpr int SWP = (SW + 3) & ~3;
    constexpr int NT = 2 * R + 1;
    constexpr int PY = TH / 4;
    __shared__ __attribute__((aligned(16))) float s_src[SH * SWP];
    __shared__ __attribute__((aligned(16))) float s_mid[SH * TW];

    const int tid = threadIdx.x;
    // XCD-aware remap of the linear workgroup id
    const unsigned nblk = gridDim.x;
    const unsigned chunk = nblk >> 3;
    unsigned lin = blockIdx.x;
    if (lin < (chunk << 3)) lin = (lin & 7u) * chunk + (lin >> 3);
    const unsigned tiles = (unsigned)(tiles_x * tiles_y);
    const unsigned img = lin / tiles;
    const unsigned t2 = lin - img * tiles;
    const int ty = (int)(t2 / (unsigned)tiles_x);
    const int tx = (int)(t2 - (unsigned)ty * (unsigned)tiles_x);
    const int x0 = tx * TW, y0 = ty * TH;
    const size_t img_off = (size_t)img * (size_t)w * (size_t)h;
    const float* __restrict__ src = in + img_off;

    // 1. source tile -> LDS
    for (int idx = tid; idx < SH * SW; idx += 256) {
        const int ly = idx / SW;
        const int lx = idx - ly * SW;
        const int gy = reflect_clamp(y0 - R + ly, h);
        const int gx = reflect_clamp(x0 - R + lx, w);
        s_src[ly * SWP + lx] = src[(size_t)gy * (size_t)w + (size_t)gx];
    }
    __syncthreads();

    // 2. row pass
    for (int it = tid; it < SH * (TW / 4); it += 256) {
        const int ly = it >> 4;
        const int q = it & 15;
        constexpr int NV = 4 + 2 * R;
        constexpr int NV4 = (NV + 3) / 4;
        float v[NV4 * 4];
        const float4* p4 = reinterpret_cast<const float4*>(&s_src[ly * SWP + 4 * q]);
#pragma unroll
        for (int c = 0; c < NV4; ++c) {
            const float4 f = p4[c];
            v[4 * c + 0] = f.x;
            v[4 * c + 1] = f.y;
            v[4 * c + 2] = f.z;
            v[4 * c + 3] = f.w;
        }
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float k = taps[NT - 1 - t];
            a0 += k * v[t];
            a1 += k * v[t + 1];
            a2 += k * v[t + 2];
            a3 += k * v[t + 3];
        }
        *reinterpret_cast<float4*>(&s_mid[ly * TW + 4 * q]) = make_float4(a0, a1, a2, a3);
    }
    __syncthreads();

    // 3. column pass
    const int c = tid & 63;
    const int g = tid >> 6;
    float acc[PY];
#pragma unroll
    for (int i = 0; i < PY; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int t = 0; t < PY + 2 * R; ++t) {
        const float v = s_mid[(g * PY + t) * TW + c];
#pragma unroll
        for (int i = 0; i < PY; ++i) {
            if (t - i >= 0 && t - i <= 2 * R) acc[i] += taps[NT - 1 - (t - i)] * v;
        }
    }
    // 4. stores
    const int x = x0 + c;
    if (x < w) {
#pragma unroll
        for (int i = 0; i < PY; ++i) {
            const int y = y0 + g * PY + i;
            if (y < h) {
                const size_t o = img_off + (size_t)y * (size_t)w + (size_t)x;
                out[o] = acc[i];
                if (DOG) {
                    const float prev = s_src[(R + g * PY + i) * SWP + R + c];
                    const float dif = acc[i] - prev;
                    dog[o] = 128.0f + dif;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Two-pass fallback for any radius (r = 0 and r > kMaxRadiusFused included): the reference's own
// structure, X pass into tmp then Y pass.  Taps in LDS.
// ---------------------------------------------------------------------------------------------
__global__ void gauss_row_generic(const float* __restrict__ in, float* __restrict__ out, int w, int h,
                                  const float* __restrict__ taps, int r) {
    extern __shared__ float s_tap[];
    for (int i = threadIdx.x; i < 2 * r + 1; i += blockDim.x) s_tap[i] = taps[i];
    __syncthreads();
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const size_t row = ((size_t)blockIdx.z * (size_t)h + (size_t)y) * (size_t)w;
    float sum = 0.0f;
    for (int t = 0; t <= 2 * r; ++t) {
        const int q = reflect_clamp(x - r + t, w);
        sum += s_tap[2 * r - t] * in[row + q];
    }
    out[row + x] = sum;
}

template <bool DOG>
__global__ void gauss_col_generic(const float* __restrict__ tmp, const float* __restrict__ prev,
                                  float* __restrict__ out, float* __restrict__ dog, int w, int h,
                                  const float* __restrict__ taps, int r) {
    extern __shared__ float s_tap[];
    for (int i = threadIdx.x; i < 2 * r + 1; i += blockDim.x) s_tap[i] = taps[i];
    __syncthreads();
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const size_t base = (size_t)blockIdx.z * (size_t)h * (size_t)w;
    float sum = 0.0f;
    for (int t = 0; t <= 2 * r; ++t) {
        const int q = reflect_clamp(y - r + t, h);
        sum += s_tap[2 * r - t] * tmp[base + (size_t)q * (size_t)w + x];
    }
    const size_t o = base + (size_t)y * (size_t)w + x;
    out[o] = sum;
    if (DOG) {
        const float dif = sum - prev[o];
        dog[o] = 128.0f + dif;
    }
}

// resizeImageNoInterpolation with host-built index maps (accumulated-double rule, Vigra
// resizeLineNoInterpolation): dst(i, j) = src(lutx[i], luty[j]).
__global__ void resample_lut_kernel(const float* __restrict__ src, float* __restrict__ dst, int ws, int hs,
                                    int wd, int hd, const int* __restrict__ lutx,
                                    const int* __restrict__ luty) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= wd) return;
    const size_t so = (size_t)blockIdx.z * (size_t)ws * (size_t)hs;
    const size_t dofs = (size_t)blockIdx.z * (size_t)wd * (size_t)hd;
    dst[dofs + (size_t)y * (size_t)wd + x] = src[so + (size_t)luty[y] * (size_t)ws + (size_t)lutx[x]];
}

// alg::dog (algorithms.cpp:52-64) as a standalone operator
__global__ void dog_kernel(const float* __restrict__ lower, const float* __restrict__ higher,
                           float* __restrict__ out, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float dif = higher[i] - lower[i];
    out[i] = 128.0f + dif;
}

template <int R>
static void launch_fused_r(hipStream_t s, const float* in, float* out, float* dog, int w, int h, int n,
                           const float* d_taps) {
    constexpr int TH = 64;
    const int tiles_x = (w + 63) / 64, tiles_y = (h + TH - 1) / TH;
    const unsigned grid = (unsigned)(tiles_x * tiles_y * n);
    if (dog)
        hipLaunchKernelGGL((blur_fused_kernel<R, TH, true>), dim3(grid), dim3(256), 0, s, in, out, dog, w, h,
                           tiles_x, tiles_y, d_taps);
    else
        hipLaunchKernelGGL((blur_fused_kernel<R, TH, false>), dim3(grid), dim3(256), 0, s, in, out, dog, w, h,
                           tiles_x, tiles_y, d_taps);
}

#define SIFT_FUSED_CASE(R) \
    case R:                \
        launch_fused_r<R>(s, in, out, dog, w, h, n, d_taps); \
        return;

void launch_blur(hipStream_t s, bool fused, const float* in, float* tmp, float* out, float* dog, int w,
                 int h, int n, const float* d_taps, int radius) {
    if (fused && radius >= 1 && radius <= kMaxRadiusFused) {
        switch (radius) {
            SIFT_FUSED_CASE(1) SIFT_FUSED_CASE(2) SIFT_FUSED_CASE(3) SIFT_FUSED_CASE(4)
            SIFT_FUSED_CASE(5) SIFT_FUSED_CASE(6) SIFT_FUSED_CASE(7) SIFT_FUSED_CASE(8)
            SIFT_FUSED_CASE(9) SIFT_FUSED_CASE(10) SIFT_FUSED_CASE(11) SIFT_FUSED_CASE(12)
            SIFT_FUSED_CASE(13) SIFT_FUSED_CASE(14) SIFT_FUSED_CASE(15) SIFT_FUSED_CASE(16)
            SIFT_FUSED_CASE(17) SIFT_FUSED_CASE(18) SIFT_FUSED_CASE(19) SIFT_FUSED_CASE(20)
            SIFT_FUSED_CASE(21) SIFT_FUSED_CASE(22) SIFT_FUSED_CASE(23) SIFT_FUSED_CASE(24)
            SIFT_FUSED_CASE(25) SIFT_FUSED_CASE(26) SIFT_FUSED_CASE(27) SIFT_FUSED_CASE(28)
            SIFT_FUSED_CASE(29) SIFT_FUSED_CASE(30) SIFT_FUSED_CASE(31) SIFT_FUSED_CASE(32)
        }
    }
    const dim3 grid((unsigned)((w + 255) / 256), (unsigned)h, (unsigned)n);
    const size_t shm = sizeof(float) * (size_t)(2 * radius + 1);
    hipLaunchKernelGGL(gauss_row_generic, grid, dim3(256), shm, s, in, tmp, w, h, d_taps, radius);
    if (dog)
        hipLaunchKernelGGL(gauss_col_generic<true>, grid, dim3(256), shm, s, (const float*)tmp, in, out, dog, w,
                           h, d_taps, radius);
    else
        hipLaunchKernelGGL(gauss_col_generic<false>, grid, dim3(256), shm, s, (const float*)tmp, in, out, dog,
                           w, h, d_taps, radius);
}

void launch_resample(hipStream_t s, const float* src, float* dst, int ws, int hs, int wd, int hd, int n,
                     const int* d_lutx, const int* d_luty) {
    const dim3 grid((unsigned)((wd + 255) / 256), (unsigned)hd, (unsigned)n);
    hipLaunchKernelGGL(resample_lut_kernel, grid, dim3(256), 0, s, src, dst, ws, hs, wd, hd, d_lutx, d_luty);
}

void launch_dog(hipStream_t s, const float* lower, const float* higher, float* out, size_t count) {
    const unsigned grid = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(dog_kernel, dim3(grid), dim3(256), 0, s, lower, higher, out, count);
}

}  // namespace sift_hip
