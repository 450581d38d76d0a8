// Descriptor stage: 4x4 cells x 8 orientation bins per keypoint.
//
// Replaces Sift::_createDecriptors (/root/reference/sift.cpp:60-110) with
// alg::orientationHistogram8 (/root/reference/algorithms.cpp:135-150), alg::normalizeVector
// (:210-223) and the discarded Sift::_eliminateVectorThreshold (sift.cpp:113-128).
//
// The reference mutates its orientation / magnitude pyramids IN PLACE through views, keypoint
// after keypoint in vector order (sift.cpp:80-92): every pixel of a keypoint's 16x16 window gets
// `ori += p.orientation` and `mag += W(lx, ly)` where W is the top-left 16x16 of
// convolveWithGauss(level, 1.6) indexed by WINDOW-LOCAL coordinates.  A later keypoint whose
// window overlaps sees the accumulated values, so each pixel carries an order-dependent float
// chain.  This kernel reproduces those chains exactly and in parallel:
//   * one workgroup owns a 64x64 core tile of a level and keeps the orientation / magnitude
//     values of the 80x80 extended tile (core + 8 px fringe) in LDS;
//   * it walks ALL keypoints of the image in vector order (ballot-compacted to those whose window
//     touches the extended tile) and applies each one's update to the pixels it holds — every
//     pixel therefore sees exactly the reference's sequence of float additions;
//   * for keypoints whose location lies in the core tile the whole window is resident, so the
//     16 cell histograms are taken right after that keypoint's update, as the reference does.
// Pixels in the fringe are updated redundantly by the neighbouring tiles; nothing is written back
// to HBM except the descriptors (the reference's mutated pyramids are private state).
#include "common.h"

#pragma clang fp contract(off)

namespace sift_hip {

__device__ __forceinline__ unsigned f32_to_u16_x86_d(float v) {
    int i;
    if (v > -2147483904.0f && v < 2147483648.0f)
        i = (int)v;
    else
        i = (int)0x80000000;
    return (unsigned)i & 0xffffu;
}

__device__ __forceinline__ int reflect_idx(int p, int n) {
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * (n - 1) - p : p;
    return p < 0 ? 0 : (p >= n ? n - 1 : p);
}

// W = top-left 16x16 of convolveWithGauss(level, 1.6f) (sift.cpp:87), one workgroup per image.
// X pass value (rounded to float) recomputed per Y tap; same operation order as the full blur.
__global__ __launch_bounds__(256) void w16_kernel(const float* __restrict__ level, float* __restrict__ w16,
                                                  int w, int h, const float* __restrict__ taps, int r) {
    const int img = blockIdx.x;
    const float* __restrict__ src = level + (size_t)img * (size_t)w * (size_t)h;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    float out = 0.0f;
    if (lx < w && ly < h) {
        float sum = 0.0f;
        for (int t = 0; t <= 2 * r; ++t) {
            const int yy = reflect_idx(ly - r + t, h);
            float row = 0.0f;
            for (int s = 0; s <= 2 * r; ++s) {
                const int xx = reflect_idx(lx - r + s, w);
                row += taps[2 * r - s] * src[(size_t)yy * (size_t)w + (size_t)xx];
            }
            sum += taps[2 * r - t] * row;
        }
        out = sum;
    }
    w16[(size_t)img * 256 + (size_t)(lx + 16 * ly)] = out;
}

constexpr int kCore = 64;
constexpr int kExt = kCore + 2 * kRegion;  // 80
constexpr int kListCap = 1024;

__global__ __launch_bounds__(256) void descriptor_kernel(const DevPlan* __restrict__ plan, int level,
                                                         const FinalKp* __restrict__ finals,
                                                         const int* __restrict__ final_cnt, int final_cap,
                                                         const long long* __restrict__ out_base,
                                                         sift_hip_keypoint* __restrict__ kp_out,
                                                         float* __restrict__ desc_out) {
    __shared__ float s_ori[kExt * kExt];
    __shared__ float s_mag[kExt * kExt];
    __shared__ float s_w16[256];
    __shared__ unsigned short s_list[kListCap];
    __shared__ float s_val[256];
    __shared__ unsigned char s_bin[256];
    __shared__ float s_hist[128];
    __shared__ float s_len[16];
    __shared__ int s_wcnt[4];
    __shared__ int s_n;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int img = blockIdx.z;
    const int D = plan->dogs;
    const int oct = level / (D + 1);
    const int w = plan->w[oct], h = plan->h[oct];
    const int cx0 = blockIdx.x * kCore, cy0 = blockIdx.y * kCore;
    const int ex0 = cx0 - kRegion, ey0 = cy0 - kRegion;
    const size_t img_off = (size_t)img * (size_t)w * (size_t)h;
    const float* __restrict__ gm = plan->mag[level] + img_off;
    const float* __restrict__ go = plan->ori[level] + img_off;
    const float* __restrict__ gg = plan->gauss[level] + img_off;
    const FinalKp* __restrict__ fin = finals + (size_t)img * (size_t)final_cap;
    const int K = final_cnt[img];
    const long long obase = out_base[img];

    // initial gradient values of the extended tile
    for (int idx = tid; idx < kExt * kExt; idx += 256) {
        const int ly = idx / kExt, lx = idx - ly * kExt;
        const int X = ex0 + lx, Y = ey0 + ly;
        const bool ok = X >= 0 && X < w && Y >= 0 && Y < h;
        const size_t o = (size_t)(ok ? Y : 0) * (size_t)w + (size_t)(ok ? X : 0);
        s_ori[idx] = ok ? go[o] : 0.0f;
        s_mag[idx] = ok ? gm[o] : 0.0f;
    }
    s_w16[tid] = plan->w16[level][(size_t)img * 256 + tid];
    __syncthreads();

    const int wlx = tid & 15, wly = tid >> 4;  // this thread's pixel inside a keypoint window

    for (int k0 = 0; k0 < K;) {
        // ---- ordered, ballot-compacted list of the next keypoints that touch this tile ----------
        if (tid == 0) s_n = 0;
        __syncthreads();
        int k_next = k0;
        for (; k_next < K; k_next += 256) {
            const int n_before = s_n;
            if (n_before + 256 > kListCap) break;  // list full: process it, then resume the scan here
            const int k = k_next + tid;
            bool hit = false;
            if (k < K) {
                const FinalKp f = fin[k];
                const int l = f.octave * D + f.index;
                hit = plan->nearest_level[l] == level && (int)f.x + kRegion > ex0 &&
                      (int)f.x - kRegion < ex0 + kExt && (int)f.y + kRegion > ey0 &&
                      (int)f.y - kRegion < ey0 + kExt;
            }
            const unsigned long long m = __ballot(hit);
            if (lane == 0) s_wcnt[wv] = __popcll(m);
            __syncthreads();
            int off = n_before;
            for (int q = 0; q < wv; ++q) off += s_wcnt[q];
            if (hit) s_list[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(k - k0);
            __syncthreads();
            if (tid == 0) s_n = n_before + s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
            __syncthreads();
        }
        const int n_list = s_n;

        // ---- walk the list in vector order --------------------------------------------------------
        for (int e = 0; e < n_list; ++e) {
            const int k = k0 + (int)s_list[e];
            const FinalKp f = fin[k];
            const int kx = f.x, ky = f.y;
            // sift.cpp:65-70 (never newly true after the orientation stage's stricter test)
            const bool kfilt = kx < kRegion || kx > w - kRegion || ky < kRegion || ky > h - kRegion;
            const bool owned = kx >= cx0 && kx < cx0 + kCore && ky >= cy0 && ky < cy0 + kCore;
            if (!kfilt) {
                const int X = kx - kRegion + wlx, Y = ky - kRegion + wly;
                const int ex = X - ex0, ey = Y - ey0;
                const bool inside = ex >= 0 && ex < kExt && ey >= 0 && ey < kExt;
                float o = 0.0f, mg = 0.0f;
                if (inside) {
                    const int idx = ey * kExt + ex;
                    o = s_ori[idx] + f.orientation;   // sift.cpp:82
                    s_ori[idx] = o;
                    mg = s_mag[idx] + s_w16[tid];     // sift.cpp:90, weighting(x, y) window-local
                    s_mag[idx] = mg;
                }
                if (owned) {
                    // alg::orientationHistogram8 inputs, staged in descriptor order:
                    // cell = (x/4)*4 + y/4 (x outer, sift.cpp:95-96), inside a cell x outer, y inner
                    const float sum = mg * gg[(size_t)Y * (size_t)w + (size_t)X];
                    unsigned i = f32_to_u16_x86_d(__builtin_floorf(o / 45.0f));
                    i = i % 7u;
                    const int slot = ((wlx >> 2) * 4 + (wly >> 2)) * 16 + (wlx & 3) * 4 + (wly & 3);
                    s_val[slot] = sum;
                    s_bin[slot] = (unsigned char)i;
                }
            }
            __syncthreads();
            if (owned) {
                const long long ok = obase + k;
                if (!kfilt) {
                    if (tid < 128) {
                        const int cell = tid >> 3, b = tid & 7;
                        float acc = 0.0f;
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const float v = s_val[cell * 16 + q];
                            acc = (s_bin[cell * 16 + q] == b) ? acc + v : acc;
                        }
                        s_hist[tid] = acc;
                    }
                    __syncthreads();
                    if (tid < 16) {  // alg::normalizeVector: length = sum of the 8 bins, sequential
                        float length = 0.0f;
#pragma unroll
                        for (int b = 0; b < 8; ++b) length += s_hist[tid * 8 + b];
                        s_len[tid] = length;
                    }
                    __syncthreads();
                    if (tid < 128) {
                        const float length = s_len[tid >> 3];
                        const float v = s_hist[tid];
                        desc_out[(size_t)ok * 128 + tid] = (length == 0.0f) ? v : v / length;
                    }
                } else if (tid < 128) {
                    desc_out[(size_t)ok * 128 + tid] = 0.0f;
                }
                if (tid == 0) {
                    sift_hip_keypoint r;
                    r.scale = plan->dog_scale[f.octave * D + f.index];
                    r.orientation = f.orientation;
                    r.x = f.x;
                    r.y = f.y;
                    r.octave = f.octave;
                    r.index = f.index;
                    r.filtered = kfilt ? 1 : 0;
                    r.has_descriptor = kfilt ? 0 : 1;
                    r.reserved = 0;
                    kp_out[ok] = r;
                }
                __syncthreads();
            }
        }
        k0 = k_next;
    }
}

void launch_w16(hipStream_t s, const DevPlan& plan, int level, const float* d_taps16, int radius16) {
    const int oct = level / (plan.dogs + 1);
    hipLaunchKernelGGL(w16_kernel, dim3((unsigned)plan.n_images), dim3(256), 0, s,
                       (const float*)plan.gauss[level], plan.w16[level], plan.w[oct], plan.h[oct], d_taps16,
                       radius16);
}

void launch_descriptors(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan, int level,
                        const FinalKp* d_final, const int* d_final_cnt, int final_cap,
                        const long long* d_out_base, sift_hip_keypoint* d_kp_out, float* d_desc_out) {
    const int oct = level / (plan.dogs + 1);
    const dim3 grid((unsigned)((plan.w[oct] + kCore - 1) / kCore), (unsigned)((plan.h[oct] + kCore - 1) / kCore),
                    (unsigned)plan.n_images);
    hipLaunchKernelGGL(descriptor_kernel, grid, dim3(256), 0, s, d_plan, level, d_final, d_final_cnt, final_cap,
                       d_out_base, d_kp_out, d_desc_out);
}

}  // namespace sift_hip
