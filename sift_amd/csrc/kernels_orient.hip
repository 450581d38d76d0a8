// Gradient maps and orientation assignment.
//
// Replaces Sift::_createMagnitudePyramid / _createOrientationPyramid
// (/root/reference/sift.cpp:130-160 with alg::gradientMagnitude / gradientOrientation,
// /root/reference/algorithms.cpp:108-116) and Sift::_orientationAssignment (sift.cpp:163-203) with
// alg::orientationHistogram36 (algorithms.cpp:118-133), Sift::_findPeaks (sift.cpp:220-286) and
// alg::vertexParabola (algorithms.cpp:153-178).
//
// Reference quirks kept: orientation is atan2f's RADIANS pushed through fmod(r + 360.f, 360.)
// (so every sample lands in histogram bin 0), the bin index is (u16)floor(ori/10) % 35, the
// histogram sum is sequential in (x outer, y inner) order, the peak set has std::set<float>
// semantics and `orientation = *peaks.begin()`.
#include "common.h"
#include "linalg3.h"

#pragma clang fp contract(off)

namespace sift_hip {

// (u16) of a float as the reference's x86-64 code does it: cvttss2si (out of range / NaN ->
// 0x80000000), then the low 16 bits.
__device__ __forceinline__ unsigned f32_to_u16_x86(float v) {
    int i;
    if (v > -2147483904.0f && v < 2147483648.0f)
        i = (int)v;
    else
        i = (int)0x80000000;
    return (unsigned)i & 0xffffu;
}

// sift.cpp:130-160: interior pixels only, border stays 0.
__global__ __launch_bounds__(256) void gradient_kernel(const float* __restrict__ g, float* __restrict__ mag,
                                                       float* __restrict__ ori, int w, int h) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const size_t base = (size_t)blockIdx.z * (size_t)w * (size_t)h;
    const size_t o = base + (size_t)y * (size_t)w + (size_t)x;
    float m = 0.0f, a = 0.0f;
    if (x >= 1 && x <= w - 2 && y >= 1 && y <= h - 2) {
        const float dx = g[o + 1] - g[o - 1];
        const float dy = g[o + (size_t)w] - g[o - (size_t)w];
        // std::sqrt(std::pow(dx, 2) + std::pow(dy, 2)) evaluated in double (algorithms.cpp:109-110)
        m = (float)__builtin_sqrt((double)dx * (double)dx + (double)dy * (double)dy);
        // std::fmod(atan2f(dy, dx) + 360.f, 360.) (algorithms.cpp:114-115); atan2f in [-pi, pi] so the
        // double fmod reduces to one conditional subtraction, which is exact.
        const float r = fdlibm_atan2f(dy, dx);
        const double s = (double)(r + 360.0f);
        a = (float)(s >= 360.0 ? s - 360.0 : s);
    }
    mag[o] = m;
    ori[o] = a;
}

// One wave per keypoint.  The 16x16 window of orientation / magnitude / Gaussian is fetched with
// row-coalesced loads, the 256 (bin, magnitude*gauss) pairs are staged in LDS in the reference's
// summation order (x outer, y inner), lane b < 36 then accumulates bin b sequentially in that
// order, and lane 0 runs the short serial peak search.
__global__ __launch_bounds__(256) void orientation_kernel(const DevPlan* __restrict__ plan,
                                                          const Candidate* __restrict__ cands,
                                                          const uint32_t* __restrict__ list,
                                                          const int* __restrict__ list_cnt, int list_cap,
                                                          OrientOut* __restrict__ out,
                                                          float* __restrict__ peaks_out) {
    __shared__ __attribute__((aligned(16))) float s_prod[4][256];
    __shared__ __attribute__((aligned(16))) unsigned char s_bin[4][256];
    __shared__ float s_hist[4][36];
    __shared__ float s_only[4][36];
    __shared__ float s_set[4][36];

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int img = blockIdx.y;
    const int cnt = list_cnt[img];
    const int D = plan->dogs;
    // the survivor count lives on the device: a fixed grid strides over the groups of 4 keypoints
    for (int grp = blockIdx.x; grp * 4 < cnt; grp += gridDim.x) {
    const int kp = grp * 4 + wv;
    const bool active = kp < cnt;

    int x = 0, y = 0, lvl = 0, w = 1, h = 1, l = 0;
    bool border = true;
    if (active) {
        const Candidate cd = cands[(size_t)img * (size_t)plan->cand_capacity + list[(size_t)img * list_cap + kp]];
        x = cd.x;
        y = cd.y;
        l = cd.octave * D + cd.index;
        lvl = plan->nearest_level[l];
        const int no = lvl / (D + 1);
        w = plan->w[no];
        h = plan->h[no];
        border = (x < kRegion || x >= w - kRegion) || (y < kRegion || y >= h - kRegion);  // sift.cpp:173-174
    }
    const bool run = active && !border;
    const int throws = run ? plan->dead_blur_radius[l] : 0;  // sift.cpp:184 (0 ok, else error code)

    if (run && throws == 0) {
        const size_t img_off = (size_t)img * (size_t)w * (size_t)h;
        const float* __restrict__ gm = plan->mag[lvl] + img_off;
        const float* __restrict__ go = plan->ori[lvl] + img_off;
        const float* __restrict__ gg = plan->gauss[lvl] + img_off;
        const int x0 = x - kRegion, y0 = y - kRegion;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int ly = it * 4 + (lane >> 4);
            const int lx = lane & 15;
            const size_t o = (size_t)(y0 + ly) * (size_t)w + (size_t)(x0 + lx);
            const float sum = gm[o] * gg[o];
            unsigned i = f32_to_u16_x86(__builtin_floorf(go[o] / 10.0f));
            i = i % 35u;
            s_prod[wv][lx * 16 + ly] = sum;
            s_bin[wv][lx * 16 + ly] = (unsigned char)i;
        }
    }
    __syncthreads();
    if (run && throws == 0 && lane < 36) {
        float acc = 0.0f;
        const float4* __restrict__ pv = reinterpret_cast<const float4*>(s_prod[wv]);
        const unsigned* __restrict__ pb = reinterpret_cast<const unsigned*>(s_bin[wv]);
        const unsigned me = (unsigned)lane;
#pragma unroll 4
        for (int q = 0; q < 64; ++q) {  // 4 samples per LDS read, still strictly in sample order
            const float4 v = pv[q];
            const unsigned b = pb[q];
            acc = ((b & 0xffu) == me) ? acc + v.x : acc;
            acc = (((b >> 8) & 0xffu) == me) ? acc + v.y : acc;
            acc = (((b >> 16) & 0xffu) == me) ? acc + v.z : acc;
            acc = ((b >> 24) == me) ? acc + v.w : acc;
        }
        s_hist[wv][lane] = acc;
        s_only[wv][lane] = acc;
    }
    __syncthreads();
    if (active && lane == 0) {
        OrientOut r;
        r.orientation = 0.0f;
        r.npeaks = 0;
        r.filtered = border ? 1 : 0;
        r.throws = (unsigned char)throws;
        if (run && throws == 0) {
            float* histo = s_hist[wv];
            float* only = s_only[wv];
            float* set = s_set[wv];
            // Sift::_findPeaks (sift.cpp:220-286)
            int max_index = 0;
            for (int i = 1; i < 36; ++i)
                if (only[max_index] < only[i]) max_index = i;  // std::max_element: first largest
            const float range = (float)((double)histo[max_index] * 0.8);
            for (int i = 0; i < 36; ++i)
                if (only[i] < range) only[i] = -1.0f;
            for (int i = 1; i < 35; ++i)
                if (only[i] < only[i - 1] || only[i] < only[i + 1]) only[i] = -1.0f;
            int n = 0;
            bool nan_first = false;
            for (int pass = 0; pass < 37; ++pass) {
                int i;
                if (pass == 0) {
                    i = max_index;
                } else {
                    i = pass - 1;
                    if (!(only[i] > -1.0f) || i == max_index) continue;
                }
                unsigned short lnx, rnx;
                float lny, rny;
                const unsigned short px = (unsigned short)(i * 10 + 5);
                const float py = histo[i];
                if (i == 0) { lnx = 355; lny = histo[35]; }
                else        { lnx = (unsigned short)((i - 1) * 10 + 5); lny = histo[i - 1]; }
                if (i == 35) { rnx = 5; rny = histo[0]; }
                else         { rnx = (unsigned short)((i + 1) * 10 + 5); rny = histo[i + 1]; }
                const float v = vertex_parabola(lnx, lny, px, py, rnx, rny);
                // std::set<float>::emplace: a NaN first element blocks every later insert (no key
                // compares less than it and it compares less than none); a later NaN is never
                // inserted; numbers insert sorted and unique.
                if (pass == 0) {
                    set[0] = v;
                    n = 1;
                    nan_first = (v != v);
                } else if (!nan_first && v == v) {
                    int pos = 0;
                    while (pos < n && set[pos] < v) ++pos;
                    if (!(pos < n && !(v < set[pos]))) {  // not equivalent to an existing key
                        for (int j = n; j > pos; --j) set[j] = set[j - 1];
                        set[pos] = v;
                        ++n;
                    }
                }
            }
            r.orientation = set[0];
            r.npeaks = (unsigned short)n;
            if (n > 1) {
                float* po = peaks_out + ((size_t)img * (size_t)list_cap + (size_t)kp) * 36;
                for (int j = 0; j < n; ++j) po[j] = set[j];
            }
        }
        out[(size_t)img * (size_t)list_cap + (size_t)kp] = r;
    }
    __syncthreads();
    }
}

void launch_gradient(hipStream_t s, const float* g, float* mag, float* ori, int w, int h, int n) {
    const dim3 grid((unsigned)((w + 255) / 256), (unsigned)h, (unsigned)n);
    hipLaunchKernelGGL(gradient_kernel, grid, dim3(256), 0, s, g, mag, ori, w, h);
}

void launch_orientation(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan, const Candidate* d_cands,
                        const uint32_t* d_list, const int* d_list_cnt, int list_cap, OrientOut* d_out,
                        float* d_peaks) {
    const dim3 grid(1024, (unsigned)plan.n_images);
    hipLaunchKernelGGL(orientation_kernel, grid, dim3(256), 0, s, d_plan, d_cands, d_list, d_list_cnt, list_cap,
                       d_out, d_peaks);
}

}  // namespace sift_hip
