// Scale-space extrema scan and edge-response filter.
//
// Replaces Sift::_findScaleSpaceExtrema (/root/reference/sift.cpp:348-379) and
// Sift::_eliminateEdgeResponses (sift.cpp:288-346, with alg::foDerivative / soDerivative,
// /root/reference/algorithms.cpp:66-106, and Vigra's inverse / linearSolve).
//
// Reference semantics kept exactly:
//  * a pixel (x, y) of a middle DoG i is a candidate iff its value is >= all, or <= all, of the 12
//    values {x-1,x} x {y-1,y} x {i-1,i,i+1} (half-open subarray, strict any() => non-strict test);
//  * candidates are emitted in the order octave, dog, x (outer), y (inner).  The scan builds one
//    64-bit mask per (column x, 64-row block) with a wavefront __ballot, counts with popcount, an
//    exclusive scan over the words in that same (octave, dog, x, y-block) order yields every
//    candidate's final position, and the expansion writes them there — no sort, no atomics.
#include <float.h>

#include "common.h"
#include "linalg3.h"

#pragma clang fp contract(off)

namespace sift_hip {

// ---------------------------------------------------------------------------------------------
// Mask kernel: one workgroup = 64 rows x 64 columns of one image of one scanned DoG level.
// The three DoG tiles (65 x 65 with the x-1 / y-1 fringe) are staged in LDS with row-coalesced
// loads; each wave then walks 16 columns with lane = row, so one __ballot per column is the
// column's candidate mask in y order.
// ---------------------------------------------------------------------------------------------
constexpr int kExTile = 64;
constexpr int kExStride = kExTile + 1;  // 65 floats: lanes (rows) hit distinct banks

__global__ __launch_bounds__(256) void extrema_mask_kernel(const float* __restrict__ d0,
                                                           const float* __restrict__ d1,
                                                           const float* __restrict__ d2, int w, int h,
                                                           int nyb, int word_base, int words_per_image,
                                                           unsigned long long* __restrict__ masks,
                                                           int* __restrict__ counts) {
    __shared__ float s[3][kExStride * kExStride];
    const int tid = threadIdx.x;
    const int xa = blockIdx.x * kExTile;  // first column of the tile
    const int yb = blockIdx.y;
    const int ya = yb * kExTile;
    const int img = blockIdx.z;
    const size_t img_off = (size_t)img * (size_t)w * (size_t)h;
    const float* __restrict__ src[3] = {d0 + img_off, d1 + img_off, d2 + img_off};

    for (int idx = tid; idx < kExStride * kExStride; idx += 256) {
        const int ly = idx / kExStride;
        const int lx = idx - ly * kExStride;
        const int gx = xa - 1 + lx, gy = ya - 1 + ly;
        const bool ok = gx >= 0 && gx < w && gy >= 0 && gy < h;
        const size_t o = (size_t)(ok ? gy : 0) * (size_t)w + (size_t)(ok ? gx : 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k][idx] = ok ? src[k][o] : 0.0f;
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ly = lane + 1;
    const int y = ya + lane;
    const bool y_ok = y >= 1 && y <= h - 2;
    // values of the previous column (x-1): [image][0 = row y-1, 1 = row y]
    float pv[3][2];
    {
        const int lx0 = wave * 16;  // column left of this wave's first column
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pv[k][0] = s[k][(ly - 1) * kExStride + lx0];
            pv[k][1] = s[k][ly * kExStride + lx0];
        }
    }
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const int lx = wave * 16 + j + 1;
        const int x = xa + lx - 1;
        float cv[3][2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            cv[k][0] = s[k][(ly - 1) * kExStride + lx];
            cv[k][1] = s[k][ly * kExStride + lx];
        }
        const float c = cv[1][1];
        bool any_gt = false, any_lt = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                any_gt = any_gt || (pv[k][r] > c) || (cv[k][r] > c);
                any_lt = any_lt || (pv[k][r] < c) || (cv[k][r] < c);
            }
        }
        const bool cand = y_ok && x >= 1 && x <= w - 2 && (!any_gt || !any_lt);
        const unsigned long long m = __ballot(cand);
        if (lane == 0 && x < w) {
            const size_t wi = (size_t)img * (size_t)words_per_image + (size_t)word_base +
                              (size_t)x * (size_t)nyb + (size_t)yb;
            masks[wi] = m;
            counts[wi] = __popcll(m);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pv[k][0] = cv[k][0];
            pv[k][1] = cv[k][1];
        }
    }
}

// Exclusive scan of one image's per-word counts (in place) + total.  One 1024-thread workgroup
// per image; each thread owns a contiguous chunk so the order is the word order.
__global__ __launch_bounds__(1024) void extrema_scan_kernel(int* __restrict__ counts, int words_per_image,
                                                            int* __restrict__ totals) {
    __shared__ int s_part[1024];
    const int img = blockIdx.x;
    int* __restrict__ c = counts + (size_t)img * (size_t)words_per_image;
    const int tid = threadIdx.x;
    const int chunk = (words_per_image + 1023) / 1024;
    const int lo = tid * chunk;
    const int hi = min(lo + chunk, words_per_image);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += c[i];
    s_part[tid] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 partial sums
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (tid >= off) ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;  // exclusive prefix of this chunk
    for (int i = lo; i < hi; ++i) {
        const int v = c[i];
        c[i] = run;
        run += v;
    }
    if (tid == 1023) totals[img] = s_part[1023];
}

// Expansion: one wave per mask word, set bits -> Candidate records at offset + rank.
__global__ __launch_bounds__(256) void extrema_expand_kernel(const DevPlan* __restrict__ plan,
                                                             const unsigned long long* __restrict__ masks,
                                                             const int* __restrict__ offsets,
                                                             Candidate* __restrict__ cands) {
    const int lane = threadIdx.x & 63;
    const int words = plan->words_per_image;
    const long long total_words = (long long)words * plan->n_images;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long wi = wave0; wi < total_words; wi += nwaves) {
        const unsigned long long m = masks[wi];
        if (m == 0ull) continue;
        const int img = (int)(wi / words);
        const int lw = (int)(wi - (long long)img * words);
        int sl = 0;
        for (int k = 1; k < plan->n_scan; ++k)
            if (lw >= plan->scan_word_base[k]) sl = k;
        const int nyb = plan->scan_nyb[sl];
        const int rel = lw - plan->scan_word_base[sl];
        const int x = rel / nyb;
        const int yb = rel - x * nyb;
        if ((m >> lane) & 1ull) {
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            Candidate c;
            c.x = (uint16_t)x;
            c.y = (uint16_t)(yb * 64 + lane);
            c.octave = (uint16_t)plan->scan_octave[sl];
            c.index = (uint16_t)plan->scan_dog[sl];
            cands[(size_t)img * (size_t)plan->cand_capacity + (size_t)offsets[wi] + (size_t)rank] = c;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Body of the per-point loop of _eliminateEdgeResponses (sift.cpp:295-345).  true => filtered.
// d0/d1/d2 = dogs(octave, index-1 / index / index+1) of one image, row pitch w.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool edge_response_filtered(const float* __restrict__ d0,
                                                       const float* __restrict__ d1,
                                                       const float* __restrict__ d2, int w, int x, int y) {
    const size_t c = (size_t)y * (size_t)w + (size_t)x;
    const size_t up = c - (size_t)w, dn = c + (size_t)w;
    const float i1c = d1[c], i1l = d1[c - 1], i1r = d1[c + 1], i1u = d1[up], i1d = d1[dn];
    const float i1ul = d1[up - 1], i1ur = d1[up + 1], i1dl = d1[dn - 1], i1dr = d1[dn + 1];
    const float i0c = d0[c], i0l = d0[c - 1], i0r = d0[c + 1], i0u = d0[up], i0d = d0[dn];
    const float i2c = d2[c], i2l = d2[c - 1], i2r = d2[c + 1], i2d = d2[dn];
    // alg::foDerivative (algorithms.cpp:69-71)
    float D[3];
    D[0] = (i1l - i1r) / 2.0f;
    D[1] = (i1u - i1d) / 2.0f;
    D[2] = (i0c - i2c) / 2.0f;
    // alg::soDerivative (algorithms.cpp:82-92)
    const float dxx = i1r + i1l - 2.0f * i1c;
    const float dyy = i1d + i1u - 2.0f * i1c;
    const float dss = i2c + i0c - 2.0f * i1c;
    const float dxy = (i1dr - i1dl - i1ur + i1ul) / 2.0f;
    const float dxs = (i2r - i2l - i0r + i0l) / 2.0f;
    const float dys = (i2d - i2d - i0d + i0u) / 2.0f;  // first two terms cancel, as in the reference
    float negH[3][3];
    negH[0][0] = dxx * -1.0f; negH[1][0] = dxy * -1.0f; negH[2][0] = dxs * -1.0f;
    negH[0][1] = dxy * -1.0f; negH[1][1] = dyy * -1.0f; negH[2][1] = dys * -1.0f;
    negH[0][2] = dxs * -1.0f; negH[1][2] = dys * -1.0f; negH[2][2] = dss * -1.0f;
    float inv[3][3];
    if (!inverse3(negH, inv)) return true;            // sift.cpp:306
    float ext[3] = {0.0f, 0.0f, 0.0f};
    if (!solve3<false>(inv, D, ext)) return true;     // sift.cpp:311
    if (ext[0] > 127.5f || ext[1] > 127.5f || ext[2] > 127.5f) return true;  // :317
    float fv = 0.0f;                                  // dot(deriv^T, extremum), :322
    fv += D[0] * ext[0];
    fv += D[1] * ext[1];
    fv += D[2] * ext[2];
    fv = (float)((double)fv * (0.5 + (double)i1c));   // :323
    if ((double)fv < 7.65) return true;               // :326
    const float tr = dxx + dyy;                       // :334
    const float prod = dxx * dyy;
    const float det = (float)((double)prod - (double)dxy * (double)dxy);  // :336
    if (det < 0.0f) return true;                      // :338
    const float t = 12.1f;                            // (f32)(std::pow(10 + 1, 2) / 10), :294
    if ((double)tr * (double)tr / (double)det > (double)t) return true;   // :343
    return false;
}

__global__ __launch_bounds__(256) void edge_filter_kernel(const DevPlan* __restrict__ plan,
                                                          const Candidate* __restrict__ cands,
                                                          const int* __restrict__ totals,
                                                          uint8_t* __restrict__ flags) {
    const int img = blockIdx.y;
    const int total = totals[img];
    const int D = plan->dogs;
    const size_t base = (size_t)img * (size_t)plan->cand_capacity;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
        const Candidate cd = cands[base + c];
        const int w = plan->w[cd.octave], h = plan->h[cd.octave];
        const size_t img_off = (size_t)img * (size_t)w * (size_t)h;
        const int l = cd.octave * D + cd.index;
        const bool f = edge_response_filtered(plan->dog[l - 1] + img_off, plan->dog[l] + img_off,
                                              plan->dog[l + 1] + img_off, w, cd.x, cd.y);
        flags[base + c] = f ? 1 : 0;
    }
}

// KAT entry: m points on one DoG triple
__global__ void edge_filter_points_kernel(const float* d0, const float* d1, const float* d2, int w, int h,
                                          const uint16_t* xs, const uint16_t* ys, int m, uint8_t* flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    flags[i] = edge_response_filtered(d0, d1, d2, w, xs[i], ys[i]) ? 1 : 0;
}

__global__ void vertex_parabola_kernel(const uint16_t* lnx, const float* lny, const uint16_t* px,
                                       const float* py, const uint16_t* rnx, const float* rny, int m,
                                       float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    out[i] = vertex_parabola(lnx[i], lny[i], px[i], py[i], rnx[i], rny[i]);
}

// ---- launchers -----------------------------------------------------------------------------------
void launch_extrema_mask(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan,
                         unsigned long long* d_masks, int* d_counts) {
    (void)d_plan;
    for (int k = 0; k < plan.n_scan; ++k) {
        const int o = plan.scan_octave[k], i = plan.scan_dog[k];
        const int w = plan.w[o], h = plan.h[o];
        const int l = o * plan.dogs + i;
        const dim3 grid((unsigned)((w + kExTile - 1) / kExTile), (unsigned)plan.scan_nyb[k],
                        (unsigned)plan.n_images);
        hipLaunchKernelGGL(extrema_mask_kernel, grid, dim3(256), 0, s, (const float*)plan.dog[l - 1],
                           (const float*)plan.dog[l], (const float*)plan.dog[l + 1], w, h, plan.scan_nyb[k],
                           plan.scan_word_base[k], plan.words_per_image, d_masks, d_counts);
    }
}

void launch_extrema_scan(hipStream_t s, const DevPlan& plan, int* d_counts, int* d_totals) {
    hipLaunchKernelGGL(extrema_scan_kernel, dim3((unsigned)plan.n_images), dim3(1024), 0, s, d_counts,
                       plan.words_per_image, d_totals);
}

void launch_extrema_expand(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan,
                           const unsigned long long* d_masks, const int* d_offsets, Candidate* d_cands) {
    const long long total_words = (long long)plan.words_per_image * plan.n_images;
    long long blocks = (total_words + 3) / 4;  // 4 waves per block, one word per wave per trip
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(extrema_expand_kernel, dim3((unsigned)blocks), dim3(256), 0, s, d_plan, d_masks,
                       d_offsets, d_cands);
}

void launch_edge_filter(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan, const Candidate* d_cands,
                        const int* d_totals, uint8_t* d_flags) {
    hipLaunchKernelGGL(edge_filter_kernel, dim3(256, (unsigned)plan.n_images), dim3(256), 0, s, d_plan,
                       d_cands, d_totals, d_flags);
}

void launch_edge_filter_points(hipStream_t s, const float* d0, const float* d1, const float* d2, int w, int h,
                               const uint16_t* xs, const uint16_t* ys, int m, uint8_t* flags) {
    hipLaunchKernelGGL(edge_filter_points_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d0, d1,
                       d2, w, h, xs, ys, m, flags);
}

void launch_vertex_parabola(hipStream_t s, const uint16_t* lnx, const float* lny, const uint16_t* px,
                            const float* py, const uint16_t* rnx, const float* rny, int m, float* out) {
    hipLaunchKernelGGL(vertex_parabola_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, lnx, lny,
                       px, py, rnx, rny, m, out);
}

}  // namespace sift_hip
