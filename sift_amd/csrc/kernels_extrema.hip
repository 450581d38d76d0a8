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

#include <cstring>

#include "common.h"
#include "linalg3.h"

#pragma clang fp contract(off)

namespace sift_hip {

// ---------------------------------------------------------------------------------------------
// Mask kernel.  One WAVE owns 63 output columns x one 64-row block of one image of one scanned DoG
// level: lane l holds column x = 63*group + l (lane 0 only supplies the x-1 neighbour of lane 1).
// The wave walks the 65 rows top to bottom with row-coalesced loads straight from HBM/L2 (every
// pixel is fetched once per wave, no LDS, no barrier), keeps the previous row in registers, gets
// the x-1 values with a wavefront shuffle and ORs one candidate bit per row into a 64-bit column
// mask.  Rows are unrolled four at a time so 12 independent loads are in flight per lane.
// ---------------------------------------------------------------------------------------------
constexpr int kExCols = 63;

__global__ __launch_bounds__(256) void extrema_mask_kernel(const float* __restrict__ d0,
                                                           const float* __restrict__ d1,
                                                           const float* __restrict__ d2, int w, int h,
                                                           int nyb, int ngroups, int word_base,
                                                           int words_per_image,
                                                           unsigned long long* __restrict__ masks,
                                                           int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int grp = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (grp >= ngroups) return;  // whole wave
    const int yb = blockIdx.y;
    const int img = blockIdx.z;
    const int x = grp * kExCols + lane;
    const int xc = x < w ? x : w - 1;  // clamped column for loads
    const size_t img_off = (size_t)img * (size_t)w * (size_t)h;
    const float* __restrict__ s0 = d0 + img_off + xc;
    const float* __restrict__ s1 = d1 + img_off + xc;
    const float* __restrict__ s2 = d2 + img_off + xc;
    const int ya = yb * 64;
    const bool x_ok = lane >= 1 && x >= 1 && x <= w - 2;

    auto row_off = [&](int y) { return (size_t)(y < 0 ? 0 : (y >= h ? h - 1 : y)) * (size_t)w; };
    // previous row (own column and x-1 column), starting with row ya-1
    float po0, po1, po2, pl0, pl1, pl2;
    {
        const size_t o = row_off(ya - 1);
        po0 = s0[o]; po1 = s1[o]; po2 = s2[o];
        pl0 = __shfl_up(po0, 1); pl1 = __shfl_up(po1, 1); pl2 = __shfl_up(po2, 1);
    }
    unsigned long long mask = 0ull;
#pragma unroll 1
    for (int j0 = 0; j0 < 64; j0 += 4) {
        float c0[4], c1[4], c2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t o = row_off(ya + j0 + u);
            c0[u] = s0[o];
            c1[u] = s1[o];
            c2[u] = s2[o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float l0 = __shfl_up(c0[u], 1), l1 = __shfl_up(c1[u], 1), l2 = __shfl_up(c2[u], 1);
            const float c = c1[u];
            const bool any_gt = (pl0 > c) || (po0 > c) || (l0 > c) || (c0[u] > c) || (pl1 > c) || (po1 > c) ||
                                (l1 > c) || (pl2 > c) || (po2 > c) || (l2 > c) || (c2[u] > c);
            const bool any_lt = (pl0 < c) || (po0 < c) || (l0 < c) || (c0[u] < c) || (pl1 < c) || (po1 < c) ||
                                (l1 < c) || (pl2 < c) || (po2 < c) || (l2 < c) || (c2[u] < c);
            const int y = ya + j0 + u;
            const bool cand = x_ok && y >= 1 && y <= h - 2 && (!any_gt || !any_lt);
            mask |= (unsigned long long)(cand ? 1u : 0u) << (j0 + u);
            po0 = c0[u]; po1 = c1[u]; po2 = c2[u];
            pl0 = l0; pl1 = l1; pl2 = l2;
        }
    }
    // lane 0 of group 0 is column 0 (never a candidate, its word must still be written)
    if ((lane >= 1 || grp == 0) && x < w) {
        const size_t wi = (size_t)img * (size_t)words_per_image + (size_t)word_base + (size_t)x * (size_t)nyb +
                          (size_t)yb;
        masks[wi] = mask;
        counts[wi] = __popcll(mask);
    }
}

// Exclusive scan of one image's per-word counts (in place) + total.  One workgroup per image; each thread owns a contiguous
// chunk so the order is the word order.  (1024 threads; 256 - a workgroup that finds a slot sooner beside the partner batch's
// descriptor kernel - was measured in round 5 and is slower alone and no faster in the pipeline: kernels_desc.hip, kGridThreads.)
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void extrema_scan_kernel(int* __restrict__ counts, int words_per_image,
                                                                    int* __restrict__ totals) {
    __shared__ int s_part[kScanThreads];
    const int img = blockIdx.x;
    int* __restrict__ c = counts + (size_t)img * (size_t)words_per_image;
    const int tid = threadIdx.x;
    // chunks are multiples of 4 words so that a thread streams its chunk with 16-byte loads / stores
    // (several in flight), when the image's slice of the array is 16-byte aligned
    const bool vec = (words_per_image & 3) == 0 && (reinterpret_cast<uintptr_t>(c) & 15u) == 0;
    const int chunk = ((words_per_image + kScanThreads - 1) / kScanThreads + 3) & ~3;
    const int lo = min(tid * chunk, words_per_image);
    const int hi = min(lo + chunk, words_per_image);
    int sum = 0;
    if (vec) {
        const int4* c4 = reinterpret_cast<const int4*>(c);
#pragma unroll 8
        for (int i = lo >> 2; i < hi >> 2; ++i) {
            const int4 v = c4[i];
            sum += v.x + v.y + v.z + v.w;
        }
    } else {
        for (int i = lo; i < hi; ++i) sum += c[i];
    }
    s_part[tid] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over the partial sums
    for (int off = 1; off < kScanThreads; off <<= 1) {
        const int v = (tid >= off) ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;  // exclusive prefix of this chunk
    if (vec) {
        int4* c4 = reinterpret_cast<int4*>(c);
#pragma unroll 8
        for (int i = lo >> 2; i < hi >> 2; ++i) {
            const int4 v = c4[i];
            int4 o;
            o.x = run;
            o.y = run + v.x;
            o.z = o.y + v.y;
            o.w = o.z + v.z;
            run = o.w + v.w;
            c4[i] = o;
        }
    } else {
        for (int i = lo; i < hi; ++i) {
            const int v = c[i];
            c[i] = run;
            run += v;
        }
    }
    if (tid == kScanThreads - 1) totals[img] = s_part[kScanThreads - 1];
}

// Expansion: one thread per mask word walks its set bits (y ascending) and writes the Candidate
// records at the word's offset: consecutive threads write consecutive runs of the output.
__global__ __launch_bounds__(256) void extrema_expand_kernel(const DevPlan* __restrict__ plan,
                                                             const unsigned long long* __restrict__ masks,
                                                             const int* __restrict__ offsets,
                                                             Candidate* __restrict__ cands,
                                                             const unsigned long long* __restrict__ fmasks,
                                                             uint8_t* __restrict__ flags) {
    const int words = plan->words_per_image;
    const long long total_words = (long long)words * plan->n_images;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long wi = (long long)blockIdx.x * blockDim.x + threadIdx.x; wi < total_words; wi += stride) {
        unsigned long long m = masks[wi];
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
        Candidate c;
        c.x = (uint16_t)x;
        c.octave = (uint16_t)plan->scan_octave[sl];
        c.index = (uint16_t)plan->scan_dog[sl];
        Candidate* out = cands + (size_t)img * (size_t)plan->cand_capacity + (size_t)offsets[wi];
        // with the fused scan the edge filter's verdicts arrive as a second bit word
        const unsigned long long fm = fmasks ? fmasks[wi] : 0ull;
        uint8_t* fout = flags ? flags + (size_t)img * (size_t)plan->cand_capacity + (size_t)offsets[wi] : nullptr;
        while (m) {
            const int bit = __ffsll((long long)m) - 1;
            c.y = (uint16_t)(yb * 64 + bit);
            *out++ = c;
            if (fout) *fout++ = (uint8_t)((fm >> bit) & 1ull);
            m &= m - 1ull;
        }
    }
}

// Expansion of the FUSED scan's words (round 6): one workgroup per (image, scan level, 32-column strip) - the columns of one
// tile column of extrema_edge_kernel, whose mask words are contiguous in the reference's candidate order (octave, dog, x outer,
// y inner: sift.cpp:352-373).  Where the strip's candidates start is the sum of the per-tile counts the scan kernel left for the
// tiles numbered in front of it (level, strip, 64-row block: a few thousand ints per image, L2-resident) - so the
// one-workgroup-per-image scan launch between the two kernels (rounds 1 - 5: 190 us alone for 11 MB, every lane walking a
// 176-byte chunk of its own) is gone, and so is the per-word count array.  Then the strip's words in order: popcount, block
// scan, every thread writes its word's run of records and flag bytes.  The image's last strip also leaves the image's total.
__global__ __launch_bounds__(256) void extrema_expand_tiles_kernel(const DevPlan* __restrict__ plan,
                                                                   const unsigned long long* __restrict__ masks,
                                                                   const unsigned long long* __restrict__ fmasks,
                                                                   const int* __restrict__ tile_counts,
                                                                   Candidate* __restrict__ cands, uint8_t* __restrict__ flags,
                                                                   int* __restrict__ totals) {
    __shared__ int s_part[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int spi = plan->strips_per_image;
    const int img = (int)blockIdx.x / spi, sg = (int)blockIdx.x - img * spi;
    int sl = 0;
    for (int k = 1; k < plan->n_scan; ++k)
        if (sg >= plan->scan_strip_base[k]) sl = k;
    const int strip = sg - plan->scan_strip_base[sl];
    const int nyb = plan->scan_nyb[sl];
    const int w = plan->w[plan->scan_octave[sl]];
    const int x0 = strip * kFxCols;
    const int cols = min(kFxCols, w - x0);
    // candidates in front of the strip: the tiles of the levels before, and of this level's strips before
    const int pre = plan->scan_tile_base[sl] + strip * nyb;
    const int* __restrict__ tc = tile_counts + (size_t)img * (size_t)plan->tiles_per_image;
    int sum = 0;
    for (int i = tid; i < pre; i += 256) sum += tc[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) s_part[wv] = sum;
    __syncthreads();
    int run = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
    const size_t w0 = (size_t)img * (size_t)plan->words_per_image + (size_t)plan->scan_word_base[sl] + (size_t)x0 * (size_t)nyb;
    const int nwords = cols * nyb;
    Candidate c;
    c.octave = (uint16_t)plan->scan_octave[sl];
    c.index = (uint16_t)plan->scan_dog[sl];
    const size_t cbase = (size_t)img * (size_t)plan->cand_capacity;
    for (int i0 = 0; i0 < nwords; i0 += 256) {
        const int i = i0 + tid;
        unsigned long long m = i < nwords ? masks[w0 + i] : 0ull;
        const int cnt = __popcll(m);
        int incl = cnt;   // inclusive scan over the wave, then over the four waves
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) s_part[wv] = incl;
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) before += k < wv ? s_part[k] : 0;
        const int all = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (m) {
            const int xc = i / nyb, yb = i - xc * nyb;
            c.x = (uint16_t)(x0 + xc);
            const unsigned long long fm = fmasks[w0 + i];
            size_t o = cbase + (size_t)(run + before + incl - cnt);
            while (m) {
                const int bit = __ffsll((long long)m) - 1;
                c.y = (uint16_t)(yb * 64 + bit);
                cands[o] = c;
                flags[o] = (uint8_t)((fm >> bit) & 1ull);
                ++o;
                m &= m - 1ull;
            }
        }
        run += all;
        __syncthreads();
    }
    if (sg == spi - 1 && tid == 0) totals[img] = run;
}

// ---------------------------------------------------------------------------------------------
// Body of the per-point loop of _eliminateEdgeResponses (sift.cpp:295-345).  true => filtered.
// d0/d1/d2 = dogs(octave, index-1 / index / index+1) of one image, row pitch w.
// ---------------------------------------------------------------------------------------------
struct EdgeTaps {   // the 18 DoG samples the per-point body reads
    float i1c, i1l, i1r, i1u, i1d, i1ul, i1ur, i1dl, i1dr;
    float i0c, i0l, i0r, i0u, i0d;
    float i2c, i2l, i2r, i2d;
};

// The two tests of _eliminateEdgeResponses that only need the 3x3 neighbourhood in the candidate's own DoG level
// (sift.cpp:334-343: determinant of the 2x2 Hessian negative, or trace^2 / determinant above 12.1).  Every test of the
// per-point body only ever SETS `filtered`, and none has a side effect, so the body's verdict is the OR of its tests in any
// order: a candidate these two already filter does not need the 3x3 QR solve at all.
__device__ __forceinline__ bool edge_curvature_filtered(float i1c, float i1l, float i1r, float i1u, float i1d, float i1ul, float i1ur,
                                                        float i1dl, float i1dr) {
    const float dxx = i1r + i1l - 2.0f * i1c;
    const float dyy = i1d + i1u - 2.0f * i1c;
    const float dxy = (i1dr - i1dl - i1ur + i1ul) / 2.0f;
    const float tr = dxx + dyy;
    const float prod = dxx * dyy;
    const float det = (float)((double)prod - (double)dxy * (double)dxy);
    if (det < 0.0f) return true;
    return (double)tr * (double)tr / (double)det > (double)12.1f;
}

__device__ __forceinline__ bool edge_response_core(const EdgeTaps& tp) {
    const float i1c = tp.i1c, i1l = tp.i1l, i1r = tp.i1r, i1u = tp.i1u, i1d = tp.i1d;
    const float i1ul = tp.i1ul, i1ur = tp.i1ur, i1dl = tp.i1dl, i1dr = tp.i1dr;
    const float i0c = tp.i0c, i0l = tp.i0l, i0r = tp.i0r, i0u = tp.i0u, i0d = tp.i0d;
    const float i2c = tp.i2c, i2l = tp.i2l, i2r = tp.i2r, i2d = tp.i2d;
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

__device__ __forceinline__ bool edge_response_filtered(const float* __restrict__ d0,
                                                       const float* __restrict__ d1,
                                                       const float* __restrict__ d2, int w, int x, int y) {
    const size_t c = (size_t)y * (size_t)w + (size_t)x;
    const size_t up = c - (size_t)w, dn = c + (size_t)w;
    EdgeTaps t;
    t.i1c = d1[c]; t.i1l = d1[c - 1]; t.i1r = d1[c + 1]; t.i1u = d1[up]; t.i1d = d1[dn];
    t.i1ul = d1[up - 1]; t.i1ur = d1[up + 1]; t.i1dl = d1[dn - 1]; t.i1dr = d1[dn + 1];
    t.i0c = d0[c]; t.i0l = d0[c - 1]; t.i0r = d0[c + 1]; t.i0u = d0[up]; t.i0d = d0[dn];
    t.i2c = d2[c]; t.i2l = d2[c - 1]; t.i2r = d2[c + 1]; t.i2d = d2[dn];
    return edge_response_core(t);
}

// ---------------------------------------------------------------------------------------------
// Fused scan + edge filter.  A workgroup stages a 32-column x 64-row tile (+1 halo) of the three DoG
// levels in LDS with row-coalesced loads, so neither the 12-sample extremum test nor the 18-sample
// edge-response body issues a scattered global load (a thread-per-candidate gather touches one cache
// line per lane and row: the standalone filter is bound by the address path, not by arithmetic).
//   * wave v scans rows 16v..16v+15, two rows x 32 columns per step; candidates are ballot-compacted
//     into the wave's LDS queue, then processed 64 at a time (dense lanes for the 3x3 QR body);
//   * results are the same per-(column, 64-row block) words as the mask kernel writes, plus a second
//     word of `filtered` bits; the expansion turns both into the candidate records and flag bytes in
//     the reference's octave / dog / x / y order.
// ---------------------------------------------------------------------------------------------
// (kFxCols = 32 columns per tile: common.h)
constexpr int kFxLead = 4;                      // columns staged left of the tile (16-byte aligned loads)
constexpr int kFxPitch = kFxCols + 2 * kFxLead; // 40 floats per staged row: x0-4 .. x0+35
constexpr int kFxRows = 64 + 2;
constexpr int kFxRow4 = kFxPitch / 4;

__device__ __forceinline__ void fx_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Persistent form: a workgroup walks tiles (image, 64-row block, 32-column strip) and fetches the NEXT tile's three DoG
// levels into registers before it starts on the current one, so the HBM latency of a tile hides under the scan and the QR
// bodies of the previous tile (barriers wait for LDS only: the fetch stays in flight).  Workgroups with equal
// blockIdx % 8 (observed to share an XCD) own a contiguous run of tiles, whose one-pixel halos then meet in that L2.
constexpr int kFxLoads = (kFxRows * kFxRow4 + 255) / 256;   // float4 per thread, level and tile
constexpr int kFxQ2 = 1024;   // second-stage list (a tile's 2048 pixels can all be candidates: the overflow runs its QR body in place)

// Round 5: ONE launch scans every level handed to it (FxPlan: up to kFxMaxLevels scan levels, their tiles numbered through) -
// the persistent workgroups simply walk on into the next level's tiles.  As a launch per level the three small octaves' scans
// were three more dependent launches behind octave 0's on the batch's critical chain, each waiting for slots beside the gradient
// pass and the partner batch's descriptors (0.75 ms for 0.12 ms of work in the pipelined timeline, profiles/r05_timeline_pipelined.txt).
constexpr int kFxMaxLevels = 16;
struct FxLevel {
    const float* d0;
    const float* d1;
    const float* d2;
    const float* d3;   // FROM_GAUSS: d0 .. d3 are four Gaussian levels, the three DoG tiles are formed on the way into LDS
    int w, h, nyb, word_base, tiles_x, tiles_img, tile_begin;
    int cnt_base;      // the level's first tile among the image's tiles in candidate order (DevPlan::scan_tile_base)
};
struct FxPlan {
    int n_levels, total_tiles, words_per_image;
    int tiles_per_image;   // DevPlan::tiles_per_image: stride of the per-tile counts
    FxLevel lv[kFxMaxLevels];
};

template <bool FROM_GAUSS>
__global__ __launch_bounds__(256, 4) void extrema_edge_kernel(FxPlan plan, unsigned long long* __restrict__ masks,
                                                           unsigned long long* __restrict__ fmasks,
                                                           int* __restrict__ tile_counts) {
    const int total_tiles = plan.total_tiles, words_per_image = plan.words_per_image;
    // level of a tile (tiles are numbered level after level; wave-uniform: a scalar loop over at most 16 entries)
    auto level_of = [&](int tile) {
        int L = 0;
        for (int i = 1; i < plan.n_levels; ++i) L = tile >= plan.lv[i].tile_begin ? i : L;
        return L;
    };
    __shared__ __attribute__((aligned(16))) float s_t[3][kFxRows * kFxPitch];
    __shared__ unsigned short s_q[4][16 * kFxCols];    // per wave: its candidates (column | row << 5)
    __shared__ unsigned short s_q2[kFxQ2];             // candidates the curvature tests let through: the QR bodies' work list
    __shared__ unsigned long long s_cm[kFxCols];       // candidate bits per column
    __shared__ unsigned long long s_fm[kFxCols];       // filtered bits per column
    __shared__ int s_qn[4];
    __shared__ int s_n2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int t, t_end, t_step;
    if ((gridDim.x & 7u) == 0) {
        const int chunk = (total_tiles + 7) >> 3;
        const int xcd = (int)blockIdx.x & 7;
        t = xcd * chunk + (int)(blockIdx.x >> 3);
        t_end = min(xcd * chunk + chunk, total_tiles);
        t_step = (int)(gridDim.x >> 3);
    } else {
        t = (int)blockIdx.x;
        t_end = total_tiles;
        t_step = (int)gridDim.x;
    }
    static_assert(kFxLoads == 3, "the prefetch registers are spelled out (an array here ends up in scratch memory)");
    const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 pa0 = z4, pa1 = z4, pa2 = z4, pb0 = z4, pb1 = z4, pb2 = z4, pc0 = z4, pc1 = z4, pc2 = z4;
    float4 pd0 = z4, pd1 = z4, pd2 = z4;   // FROM_GAUSS: the fourth level
    // tile: rows ya-1 .. ya+64, columns x0-4 .. x0+35 as 16-byte groups (w is a multiple of 4 here).  Rows
    // and groups outside the image are clamped; clamped samples are only read by non-candidates.
#define SIFT_FX_LOAD(I, A, B, C, D)                                                                  \
    {                                                                                                \
        const int e = tid + 256 * (I);                                                               \
        if (e < kFxRows * kFxRow4) {                                                                 \
            const int r = e / kFxRow4, c4 = e - r * kFxRow4;                                         \
            int gy = ya_ - 1 + r, gx = x0_ - kFxLead + 4 * c4;                                       \
            gy = gy < 0 ? 0 : (gy >= h ? h - 1 : gy);                                                \
            gx = gx < 0 ? 0 : (gx > w - 4 ? w - 4 : gx);                                             \
            const size_t o = img_off_ + (size_t)gy * (size_t)w + (size_t)gx;                         \
            A = *reinterpret_cast<const float4*>(d0 + o);                                            \
            B = *reinterpret_cast<const float4*>(d1 + o);                                            \
            C = *reinterpret_cast<const float4*>(d2 + o);                                            \
            if (FROM_GAUSS) D = *reinterpret_cast<const float4*>(d3 + o);                            \
        }                                                                                            \
    }
#define SIFT_FX_LOAD_TILE(TILE)                                                                      \
    {                                                                                                \
        const FxLevel& lv_ = plan.lv[level_of(TILE)];                                                \
        const float* __restrict__ d0 = lv_.d0;                                                       \
        const float* __restrict__ d1 = lv_.d1;                                                       \
        const float* __restrict__ d2 = lv_.d2;                                                       \
        const float* __restrict__ d3 = lv_.d3;                                                       \
        const int w = lv_.w, h = lv_.h, tl_ = (TILE) - lv_.tile_begin;                               \
        const int img_ = tl_ / lv_.tiles_img, rem_ = tl_ - img_ * lv_.tiles_img;                     \
        const int yb_ = rem_ / lv_.tiles_x, x0_ = (rem_ - yb_ * lv_.tiles_x) * kFxCols, ya_ = yb_ * 64; \
        const size_t img_off_ = (size_t)img_ * (size_t)w * (size_t)h;                                \
        SIFT_FX_LOAD(0, pa0, pb0, pc0, pd0) SIFT_FX_LOAD(1, pa1, pb1, pc1, pd1) SIFT_FX_LOAD(2, pa2, pb2, pc2, pd2) \
    }
#define SIFT_FX_DOG(H, L) make_float4(128.0f + ((H).x - (L).x), 128.0f + ((H).y - (L).y), 128.0f + ((H).z - (L).z), 128.0f + ((H).w - (L).w))
#define SIFT_FX_STORE(I, A, B, C, D)                                                                 \
    {                                                                                                \
        const int e = tid + 256 * (I);                                                               \
        if (e < kFxRows * kFxRow4) { /* r * kFxPitch + 4 * c4 == 4 * e */                            \
            if (FROM_GAUSS) {   /* alg::dog (algorithms.cpp:52-64): 128 + (higher - lower), two roundings */ \
                *reinterpret_cast<float4*>(&s_t[0][4 * e]) = SIFT_FX_DOG(B, A);                      \
                *reinterpret_cast<float4*>(&s_t[1][4 * e]) = SIFT_FX_DOG(C, B);                      \
                *reinterpret_cast<float4*>(&s_t[2][4 * e]) = SIFT_FX_DOG(D, C);                      \
            } else {                                                                                 \
                *reinterpret_cast<float4*>(&s_t[0][4 * e]) = A;                                      \
                *reinterpret_cast<float4*>(&s_t[1][4 * e]) = B;                                      \
                *reinterpret_cast<float4*>(&s_t[2][4 * e]) = C;                                      \
            }                                                                                        \
        }                                                                                            \
    }
    if (t < t_end) SIFT_FX_LOAD_TILE(t)
    while (t < t_end) {
        const FxLevel& lv = plan.lv[level_of(t)];
        const int w = lv.w, h = lv.h, nyb = lv.nyb, word_base = lv.word_base, tiles_x = lv.tiles_x, tiles_img = lv.tiles_img;
        const int tl = t - lv.tile_begin;
        const int img = tl / tiles_img, rem = tl - img * tiles_img;
        const int yb = rem / tiles_x, x0 = (rem - yb * tiles_x) * kFxCols, ya = yb * 64;
        SIFT_FX_STORE(0, pa0, pb0, pc0, pd0) SIFT_FX_STORE(1, pa1, pb1, pc1, pd1) SIFT_FX_STORE(2, pa2, pb2, pc2, pd2)
        if (tid < kFxCols) { s_fm[tid] = 0ull; s_cm[tid] = 0ull; }
        if (tid < 4) s_qn[tid] = 0;
        if (tid == 4) s_n2 = 0;
        fx_lds_barrier();
        const int tn = t + t_step;
        if (tn < t_end) SIFT_FX_LOAD_TILE(tn)   // stays in flight through the scan and the QR bodies

    // ---- scan: lane = 4 consecutive columns x 2 consecutive rows; wave v owns rows 16v .. 16v+15 -----------------------------
    // All 18 LDS reads of a lane (three tile rows of the three levels: a 16-byte group and the sample left of it) are issued
    // together; the 12-neighbour extremum test of the 8 pixels then runs on registers: pairwise max / min over {x-1, x} per
    // level and row once, three-input max / min over the rest ("no neighbour greater" == "max of the neighbours <= centre";
    // DoG samples are finite, so no NaN ordering question arises).
    {
        // Lane -> (column group cg, row pair): the eight lanes of a column group row share an LDS row (128 contiguous bytes); the
        // 16 lanes a 16-byte LDS read serves together hold two row pairs, taken FOUR rows apart (row pairs 0,2,1,3,4,6,5,7 over
        // lane >> 3): 4 rows x 40 floats = 160 floats = 32 banks past the first group's, so the two groups never meet in a bank
        // (two rows apart they overlap in half of them).  The sample left of a lane's four columns is its left neighbour's last
        // one: a DPP move, and an LDS read only for the column group at the tile's left edge (with one read per lane all 64
        // lanes would land in the 16 banks = 3 mod 4, four deep).
        const int cg = lane & 7, rs = lane >> 3;
        const int rp = (rs & 4) | ((rs & 1) << 1) | ((rs >> 1) & 1);
        const int row0 = 16 * wv + 2 * rp;                     // first of the lane's two output rows (tile-local)
        const int at0 = row0 * kFxPitch + kFxLead + 4 * cg;    // LDS index of (row0 - 1, first column): LDS row = tile row + 1
        float v[3][3][5];                                      // [level][LDS row row0 + R][column -1 .. 3]
#pragma unroll
        for (int L = 0; L < 3; ++L)
#pragma unroll
            for (int R = 0; R < 3; ++R) {
                const float4 q4 = *reinterpret_cast<const float4*>(&s_t[L][at0 + R * kFxPitch]);
                v[L][R][1] = q4.x; v[L][R][2] = q4.y; v[L][R][3] = q4.z; v[L][R][4] = q4.w;
            }
#pragma unroll
        for (int L = 0; L < 3; ++L)
#pragma unroll
            for (int R = 0; R < 3; ++R) {
                // row_shr:1 within the 16-lane DPP row: lane l receives lane l-1's value (lanes 0 and 8 of a row are column group 0)
                float left = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[L][R][4]), 0x111, 0xf, 0xf, false));
                if (cg == 0) left = s_t[L][at0 + R * kFxPitch - 1];
                v[L][R][0] = left;
            }
        float pmx[3][3][4], pmn[3][3][4];
#pragma unroll
        for (int L = 0; L < 3; ++L)
#pragma unroll
            for (int R = 0; R < 3; ++R)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pmx[L][R][i] = fmaxf(v[L][R][i], v[L][R][i + 1]);
                    pmn[L][R][i] = fminf(v[L][R][i], v[L][R][i + 1]);
                }
        unsigned bits = 0u;   // bit a * 4 + i: pixel (row0 + a, 4 cg + i) is a candidate
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int y = ya + row0 + a;
            const bool y_ok = y >= 1 && y <= h - 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x = x0 + 4 * cg + i;
                const float c = v[1][a + 1][i + 1];
                const float nmax = fmaxf(fmaxf(fmaxf(pmx[0][a][i], pmx[0][a + 1][i]), fmaxf(pmx[2][a][i], pmx[2][a + 1][i])),
                                         fmaxf(pmx[1][a][i], v[1][a + 1][i]));
                const float nmin = fminf(fminf(fminf(pmn[0][a][i], pmn[0][a + 1][i]), fminf(pmn[2][a][i], pmn[2][a + 1][i])),
                                         fminf(pmn[1][a][i], v[1][a + 1][i]));
                const bool cand = y_ok && x >= 1 && x <= w - 2 && (!(nmax > c) || !(nmin < c));
                bits |= (cand ? 1u : 0u) << (a * 4 + i);
            }
        }
        if (bits) {
            // candidate bits of the lane's four columns into the columns' 64-row words
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned two = ((bits >> i) & 1u) | (((bits >> (4 + i)) & 1u) << 1);
                if (two) atomicOr(&s_cm[4 * cg + i], (unsigned long long)two << row0);
            }
            // and the candidates themselves into the wave's queue
            int at = atomicAdd(&s_qn[wv], __popc(bits));
            unsigned rest = bits;
            while (rest) {
                const int b = __ffs((int)rest) - 1;
                rest &= rest - 1u;
                s_q[wv][at++] = (unsigned short)((4 * cg + (b & 3)) | ((row0 + (b >> 2)) << 5));
            }
        }
    }
    fx_lds_barrier();
    // ---- first stage over the candidates (the four queues walked as one list): the two curvature tests of the edge filter,
    // which most candidates fail; the others are compacted into the second list so that the 3x3 QR bodies run on full waves
    const int n0 = s_qn[0], n1 = n0 + s_qn[1], n2 = n1 + s_qn[2], n3 = n2 + s_qn[3];
    auto qr_body = [&](int qc, int qr) {
        const int at = (qr + 1) * kFxPitch + (qc + kFxLead);
        const int up = at - kFxPitch, dn = at + kFxPitch;
        EdgeTaps tp;
        tp.i1c = s_t[1][at]; tp.i1l = s_t[1][at - 1]; tp.i1r = s_t[1][at + 1]; tp.i1u = s_t[1][up]; tp.i1d = s_t[1][dn];
        tp.i1ul = s_t[1][up - 1]; tp.i1ur = s_t[1][up + 1]; tp.i1dl = s_t[1][dn - 1]; tp.i1dr = s_t[1][dn + 1];
        tp.i0c = s_t[0][at]; tp.i0l = s_t[0][at - 1]; tp.i0r = s_t[0][at + 1]; tp.i0u = s_t[0][up]; tp.i0d = s_t[0][dn];
        tp.i2c = s_t[2][at]; tp.i2l = s_t[2][at - 1]; tp.i2r = s_t[2][at + 1]; tp.i2d = s_t[2][dn];
        if (edge_response_core(tp)) atomicOr(&s_fm[qc], 1ull << qr);
    };
    for (int gb = 0; gb < n3; gb += 256) {   // block-uniform trip count: every lane takes part in the ballots
        const int g = gb + tid;
        bool keep = false;
        unsigned e = 0u;
        if (g < n3) {
            const int qi = g < n0 ? 0 : g < n1 ? 1 : g < n2 ? 2 : 3;
            const int off = qi == 0 ? 0 : qi == 1 ? n0 : qi == 2 ? n1 : n2;
            e = s_q[qi][g - off];
            const int qc = (int)(e & 31u), qr = (int)(e >> 5);
            const int at = (qr + 1) * kFxPitch + (qc + kFxLead);
            const bool curved = edge_curvature_filtered(s_t[1][at], s_t[1][at - 1], s_t[1][at + 1], s_t[1][at - kFxPitch], s_t[1][at + kFxPitch],
                                                        s_t[1][at - kFxPitch - 1], s_t[1][at - kFxPitch + 1], s_t[1][at + kFxPitch - 1],
                                                        s_t[1][at + kFxPitch + 1]);
            if (curved) atomicOr(&s_fm[qc], 1ull << qr);
            keep = !curved;
        }
        const unsigned long long bal = __ballot(keep);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_n2, __popcll(bal));
        base = __shfl(base, 0);
        if (keep) {
            const int pos = base + (int)__popcll(bal & ((1ull << lane) - 1ull));
            if (pos < kFxQ2) s_q2[pos] = (unsigned short)e;
            else qr_body((int)(e & 31u), (int)(e >> 5));   // list full (a tile of ties): in place
        }
    }
    fx_lds_barrier();
    // ---- second stage: the 3x3 QR bodies -------------------------------------------------------------------------------
    const int m2 = min(s_n2, kFxQ2);
    for (int g = tid; g < m2; g += 256) {
        const unsigned e = s_q2[g];
        qr_body((int)(e & 31u), (int)(e >> 5));
    }
    fx_lds_barrier();
    if (wv == 0) {
        int cnt = 0;
        if (tid < kFxCols && x0 + tid < w) {
            const unsigned long long m = s_cm[tid];
            const size_t wi = (size_t)img * (size_t)words_per_image + (size_t)word_base + (size_t)(x0 + tid) * (size_t)nyb +
                              (size_t)yb;
            masks[wi] = m;
            fmasks[wi] = s_fm[tid];
            cnt = __popcll(m);
        }
        // the tile's candidates: what the expansion sums over the tiles in front of a strip (no scan launch, no atomics; every
        // tile of the plan writes its count in every batch, so nothing needs clearing)
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
        if (tid == 0) tile_counts[(size_t)img * (size_t)plan.tiles_per_image + (size_t)(lv.cnt_base + (x0 / kFxCols) * nyb + yb)] = cnt;
    }
        t = tn;
    }
#undef SIFT_FX_STORE
#undef SIFT_FX_LOAD_TILE
#undef SIFT_FX_LOAD
}

// (Round 3 also built a STREAMING form of this kernel - every wave on its own strip of 248 columns x one 64-row block, three rows
// of the three levels in registers, candidates queued with their 18 samples in a per-wave LDS ring - which was bit-identical and
// used 0.64 of this kernel's wave-cycles at octave 0 alone, but took as long beside the gradient kernel and twice as long on the
// small octaves; it was an option, off, and was removed in round 4.)

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
        const int ngroups = (w + kExCols - 1) / kExCols;  // lanes 1..63 of group g cover x = 63g+1 .. 63g+63
        const dim3 grid((unsigned)((ngroups + 3) / 4), (unsigned)plan.scan_nyb[k], (unsigned)plan.n_images);
        hipLaunchKernelGGL(extrema_mask_kernel, grid, dim3(256), 0, s, (const float*)plan.dog[l - 1],
                           (const float*)plan.dog[l], (const float*)plan.dog[l + 1], w, h, plan.scan_nyb[k], ngroups,
                           plan.scan_word_base[k], plan.words_per_image, d_masks, d_counts);
    }
}

// the fused kernel stages 16-byte groups: every scanned octave's rows must be 16-byte aligned and at least one group wide
// Persistent workgroups per CU of the fused scan: 4 = all the kernel's registers allow (122 VGPRs x 4 waves per SIMD is the whole
// register file, so nothing else runs on a CU while the scan holds it).  Round 5 measured 3 and 2 (registers and LDS left for the
// partner batch's descriptor kernel): 2.71 - 2.80 ms per step every way (profiles/r05_extrema_per_cu_ab.txt); the option is gone.
constexpr int kExtremaPerCu = 4;

// rows of every scanned octave 16-byte aligned (width a multiple of 4; the level buffers are carved 256-byte aligned, which
// the launch checks for the pointers the selected variant reads)
bool extrema_edge_supported(const DevPlan& plan) {
    for (int k = 0; k < plan.n_scan; ++k) {
        const int o = plan.scan_octave[k];
        if (plan.w[o] % 4 != 0 || plan.w[o] < 4) return false;
        const int l = o * plan.dogs + plan.scan_dog[k], gl = o * (plan.dogs + 1) + plan.scan_dog[k];
        for (int j = -1; j <= 1; ++j)
            if ((uintptr_t)plan.dog[l + j] & 15u) return false;
        for (int j = -1; j <= 2; ++j)
            if ((uintptr_t)plan.gauss[gl + j] & 15u) return false;
    }
    return true;
}

void launch_extrema_edge(hipStream_t s, const DevPlan& plan, unsigned long long* d_masks, unsigned long long* d_fmasks,
                         int* d_tile_counts, bool from_gauss, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int k_end = plan.n_scan;
    // persistent workgroups, 4 per CU, with EQUAL shares of the tiles: a workgroup that finds no CU free starts when the
    // others end and doubles the launch's time
    int cap = kExtremaPerCu * resident_cus();
    cap = cap >= 8 ? (cap & ~7) : (cap > 0 ? cap : 1);
    int k = 0;
    bool k_first = true;
    while (k < k_end) {   // kFxMaxLevels scan levels per launch (the bench plan has four)
        FxPlan fp;
        std::memset(&fp, 0, sizeof(fp));
        fp.words_per_image = plan.words_per_image;
        fp.tiles_per_image = plan.tiles_per_image;
        long long total = 0;
        for (; k < k_end && fp.n_levels < kFxMaxLevels; ++k) {
            const int o = plan.scan_octave[k], i = plan.scan_dog[k];
            const int l = o * plan.dogs + i;
            FxLevel& lv = fp.lv[fp.n_levels];
            if (from_gauss) {   // DoG level j of the octave = 128 + (g[j + 1] - g[j]): the scan of (o, i) reads g(o, i-1 .. i+2)
                const int gl = o * (plan.dogs + 1) + i;
                lv.d0 = plan.gauss[gl - 1]; lv.d1 = plan.gauss[gl]; lv.d2 = plan.gauss[gl + 1]; lv.d3 = plan.gauss[gl + 2];
            } else {
                lv.d0 = plan.dog[l - 1]; lv.d1 = plan.dog[l]; lv.d2 = plan.dog[l + 1]; lv.d3 = nullptr;
            }
            lv.w = plan.w[o]; lv.h = plan.h[o]; lv.nyb = plan.scan_nyb[k]; lv.word_base = plan.scan_word_base[k];
            lv.tiles_x = (lv.w + kFxCols - 1) / kFxCols;
            lv.tiles_img = lv.tiles_x * lv.nyb;
            const long long tiles = (long long)lv.tiles_img * plan.n_images;
            if (total + tiles > 0x7fffffffLL) break;   // (tile numbers are ints)
            lv.tile_begin = (int)total;
            lv.cnt_base = plan.scan_tile_base[k];
            total += tiles;
            ++fp.n_levels;
        }
        if (fp.n_levels == 0) break;
        fp.total_tiles = (int)total;
        int grid = total < cap ? (int)total : cap;
        if (grid >= 8) grid &= ~7;
        // (timing events: a plan of more than kFxMaxLevels scan levels makes several launches; the events bracket them all)
        hipEvent_t ea = k_first ? ev_start : nullptr, eb = k >= k_end ? ev_stop : nullptr;
        k_first = false;
        if (from_gauss) hipExtLaunchKernelGGL((extrema_edge_kernel<true>), dim3((unsigned)grid), dim3(256), 0, s, ea, eb, 0, fp, d_masks, d_fmasks, d_tile_counts);
        else hipExtLaunchKernelGGL((extrema_edge_kernel<false>), dim3((unsigned)grid), dim3(256), 0, s, ea, eb, 0, fp, d_masks, d_fmasks, d_tile_counts);
    }
}

void launch_extrema_scan(hipStream_t s, const DevPlan& plan, int* d_counts, int* d_totals) {
    hipLaunchKernelGGL(extrema_scan_kernel, dim3((unsigned)plan.n_images), dim3(kScanThreads), 0, s, d_counts,
                       plan.words_per_image, d_totals);
}

void launch_extrema_expand(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan,
                           const unsigned long long* d_masks, const int* d_offsets, Candidate* d_cands,
                           const unsigned long long* d_fmasks, uint8_t* d_flags) {
    const long long total_words = (long long)plan.words_per_image * plan.n_images;
    long long blocks = (total_words + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(extrema_expand_kernel, dim3((unsigned)blocks), dim3(256), 0, s, d_plan, d_masks,
                       d_offsets, d_cands, d_fmasks, d_flags);
}

void launch_extrema_expand_tiles(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan, const unsigned long long* d_masks,
                                 const unsigned long long* d_fmasks, const int* d_tile_counts, Candidate* d_cands, uint8_t* d_flags,
                                 int* d_totals) {
    const unsigned grid = (unsigned)plan.strips_per_image * (unsigned)plan.n_images;
    hipLaunchKernelGGL(extrema_expand_tiles_kernel, dim3(grid), dim3(256), 0, s, d_plan, d_masks, d_fmasks, d_tile_counts, d_cands,
                       d_flags, d_totals);
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

// The runtime builds a translation unit's device code on the first launch of any of its kernels, and two host threads that make
// their first launches at the same time (several contexts, one thread each) were seen to crash inside that step
// (tools/asan_example.sh: SEGV below hipLaunchKernel).  sift_hip_create touches every unit once, under a lock.
__global__ void tu_probe_extrema_kernel() {}
void tu_touch_extrema(hipStream_t s) { hipLaunchKernelGGL(tu_probe_extrema_kernel, dim3(1), dim3(1), 0, s); }

}  // namespace sift_hip
