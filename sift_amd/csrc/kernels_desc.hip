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
// chain.  descriptor_wave_kernel reproduces those chains exactly and in parallel (nothing is written back to HBM
// except the descriptors: the reference's mutated pyramids are private state): one wave per keypoint recomputes
// the chains of its window's 256 pixels from the initial maps and the preceding neighbours, found through a grid
// of 16 px cells.  (A tile-per-wave form - a wave owning a 32x32 tile of keypoint locations with the tile's pixels
// in LDS - was built in round 4, lost in the pipeline and was removed in round 6: DESIGN.md section 4.)
#include <cstdio>

#include "common.h"
#include "hist_bins.h"

#pragma clang fp contract(off)

namespace sift_hip {

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

// =====================================================================================================
// Wave-per-keypoint form (the default).  (Round 1's kernel gave a workgroup a 48x48 tile with an 8 px fringe in LDS and
// walked the tile's ordered keypoint list in batches of eight: a keypoint near a tile border was processed by up to four
// workgroups and every tile paid two barriers per batch - 1.23 against 0.85 ms; removed in round 4.)  Here ONE WAVE
// owns ONE keypoint p and recomputes, for the 256 pixels of p's window, the float chains the reference's in-place updates
// build (sift.cpp:80-92): a pixel's value when p's histograms are taken is
//     initial value (+ orientation_q, + weighting(window-local position in q))   for every keypoint q that precedes p in the
//     vector AND whose 16x16 window covers the pixel, in vector order, then p's own update.
// Only keypoints within 15 px of p in x and y can cover a pixel of p's window; they are found through a uniform grid of
// 16 px cells (the 3x3 cells around p's) that holds every keypoint of the level once.  The preceding neighbours are taken
// in ascending vector index by repeated wave-wide minimum (DPP), each applied to the lane's four pixels with its fields in
// scalar registers.  Chains are short (a pixel is covered by ~2.4 windows on the bench frames), nothing is shared between
// waves, there is no barrier, no tile fringe and no redundant update; the price is that overlapping windows recompute
// each other's additions.
// Lane l owns the pixels (4*(l>>4) + i, l & 15), i = 0..3, of the window (window-local x, y): four consecutive x of one row
// (one 16-byte load per map), and the four lanes of a DPP quad hold the four rows of one 4x4 cell: the cell's 16 samples
// reach every lane of the quad by quad-broadcast in the reference's order (x outer, y inner), lane y of the quad
// accumulating bins 2y and 2y+1.  The 8 bins of a cell then sit in one quad in bin order: lane l stores floats 2l, 2l+1 of
// the descriptor, 512 contiguous bytes per wave.
// =====================================================================================================
constexpr int kCellShift = 4;            // 16 px grid cells
// Weighting table in LDS, x-major: column x holds [15 zeros][weighting(x, 0..15)][zeros], and 15 all-zero columns stand on
// either side, so that EVERY window-local coordinate a neighbour's window can put a pixel at (-15 .. 30 in x and y) reads a
// real +0.0f outside the 16x16 window: adding it changes nothing (a magnitude is never -0.0f), which spares the walk over the
// neighbours a select per pixel.  Stride 52: the four column groups of a wave (x = 0, 4, 8, 12) start 16 banks apart.
constexpr int kW16Stride = 46;   // (round 5: 8-byte entries are conflict-free at any stride; 46 = 15 + 16 + 15 keeps eight workgroups per CU)
constexpr int kW16Bias = 15 * kW16Stride + 15;
constexpr int kW16Size = 46 * kW16Stride;
// Round 5: an entry of the table is a PAIR {weighting, mask}: mask = all ones inside the 16x16 window, 0 outside.  A neighbour's
// orientation is added as `ori + (theta & mask)` - theta inside its window, +0.0f outside, and an orientation is never -0.0f
// (it starts as a value >= +0 or NaN - kernels_orient.hip - and only (-0) + (-0) makes a -0), so `+ 0.0f` leaves every value,
// NaNs included, as it was.  One 8-byte LDS read per pixel and neighbour then serves BOTH chains and the walk needs no
// coverage test at all: add, and, add - three instructions per pixel where the compare / add / select / add form took five
// and a row test per neighbour (583 -> ~455 vector instructions per keypoint; the 16 lanes an 8-byte read serves together
// hold 16 consecutive y of one column: 128 contiguous bytes, conflict-free at any column stride).
struct W16Entry { float w; unsigned mask; };

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at a 4-byte aligned address

struct DescGridLevel {
    int w, h, dogs;
    int cw, ch, cell_base, cells_per_image;   // this level's cells inside an image's cell array
    const float* mag;
    const float* ori;
    const float* gauss;
    const float* w16;
    int* wire_sums;   // multi-GPU jobs (option "wire_count"): [0] flag, [1 + slot / 64] floats the block puts on the wire (kernels_wire.hip); or null
};

// cell of a keypoint, or -1 if the descriptor stage's own bounds test fails (sift.cpp:65-70)
__device__ __forceinline__ int desc_cell_of(const DevPlan* __restrict__ plan, const FinalKp& f, int& level) {
    const int D = plan->dogs;
    level = plan->nearest_level[f.octave * D + f.index];
    const int o = level / (D + 1);
    const int w = plan->w[o], h = plan->h[o];
    const int kx = f.x, ky = f.y;
    if (kx < kRegion || kx > w - kRegion || ky < kRegion || ky > h - kRegion) return -1;
    return plan->desc_cell_base[level] + (ky >> kCellShift) * plan->desc_cw[level] + (kx >> kCellShift);
}

// The grid of one image, by one workgroup: counts per cell, exclusive scan (cell_off[0 .. cpi], the last entry is the
// total), records into their cells.  No workgroup shares an image, so the counters live in LDS while the image's cells fit
// (LDS_COUNTS) and in the image's slice of cell_cnt otherwise; after the scan a cell's counter holds its first free pool
// slot, so the fill needs one atomic per record and no second array.  A keypoint that fails the descriptor stage's own
// bounds test (sift.cpp:65-70) is emitted right here: filtered, no descriptor.
// (Round 5 measured 256 threads / 32 KB of counters - a workgroup that takes the first four wave slots that free instead of
// waiting for half an empty CU beside the gradient pass's workgroups: 2.72 - 2.77 against 2.68 - 2.73 ms per step with two batches
// in flight, 3.05 - 3.11 against 3.00 - 3.06 with one: the kernel alone is slower and the later start of the descriptor kernel it was
// meant to cure is not what bounds the phase.  Not kept; the same for extrema_scan_kernel.)
constexpr int kGridThreads = 1024;
constexpr int kGridLdsCells = 16384;   // 64 KB of counters: 1080p has 8160 cells, 4K 32400 (global counters)

template <bool LDS_COUNTS>
__global__ __launch_bounds__(kGridThreads) void desc_grid_kernel(const DevPlan* __restrict__ plan, const FinalKp* __restrict__ finals,
                                                                 const int* __restrict__ final_cnt, int final_cap,
                                                                 int* __restrict__ cell_cnt, int* __restrict__ cell_off,
                                                                 FinalKp* __restrict__ pool, int pool_cap,
                                                                 long long* out_base, int compute_base,
                                                                 sift_hip_keypoint* __restrict__ kp_out, float* __restrict__ desc_out,
                                                                 long long out_cap) {
    __shared__ int s_cnt[LDS_COUNTS ? kGridLdsCells : 1];
    __shared__ int s_part[kGridThreads];
    __shared__ long long s_wsum[kGridThreads / 64];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int K = final_cnt[img];
    // The image's first output slot = the keypoints of the images in front of it.  compute_base: summed right here (round 5: as
    // a launch of its own - out_base_kernel, one workgroup - the scan was one more dependent launch between the cleanup chain and
    // the descriptors, ~60 us waiting for a slot beside the partner batch's pyramid); else the host has uploaded the offsets.
    long long my_base;
    if (compute_base) {
        long long sum = 0;
        for (int i = tid; i < img; i += kGridThreads) sum += final_cnt[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
        if ((tid & 63) == 0) s_wsum[tid >> 6] = sum;
        __syncthreads();
        long long t = 0;
#pragma unroll
        for (int wv = 0; wv < kGridThreads / 64; ++wv) t += s_wsum[wv];
        my_base = t;
        if (tid == 0) out_base[img] = t;
    } else {
        my_base = out_base[img];
    }
    const int cpi = plan->desc_cells_per_image;
    int* cnt = LDS_COUNTS ? s_cnt : cell_cnt + (size_t)img * (size_t)(cpi + 1);
    int* off = cell_off + (size_t)img * (size_t)(cpi + 1);
    const FinalKp* __restrict__ fin = finals + (size_t)img * (size_t)final_cap;
    for (int i = tid; i < cpi; i += kGridThreads) cnt[i] = 0;
    __syncthreads();
    for (int k = tid; k < K; k += kGridThreads) {
        const FinalKp f = fin[k];
        int level;
        const int cell = desc_cell_of(plan, f, level);
        if (cell >= 0) {
            atomicAdd(&cnt[cell], 1);
        } else {
            const long long ok = my_base + k;
            if (ok < out_cap) {
                sift_hip_keypoint r;
                r.scale = plan->dog_scale[f.octave * plan->dogs + f.index];
                r.orientation = f.orientation;
                r.x = f.x; r.y = f.y; r.octave = f.octave; r.index = f.index;
                r.filtered = 1; r.has_descriptor = 0; r.reserved = 0;
                kp_out[ok] = r;
                float4* d = reinterpret_cast<float4*>(desc_out + (size_t)ok * 128);
                for (int i = 0; i < 32; ++i) d[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
    }
    __syncthreads();
    // exclusive scan: each thread owns a run of consecutive cells
    const int chunk = (cpi + kGridThreads - 1) / kGridThreads;
    const int lo = tid * chunk, hi = min(lo + chunk, cpi);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < kGridThreads; o <<= 1) {
        const int v = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
    for (int i = lo; i < hi; ++i) {
        const int v = cnt[i];
        off[i] = run;
        cnt[i] = run;   // from here on: the cell's next free pool slot
        run += v;
    }
    if (tid == kGridThreads - 1) off[cpi] = s_part[kGridThreads - 1];
    __syncthreads();
    FinalKp* pl = pool + (size_t)img * (size_t)pool_cap;
    for (int k = tid; k < K; k += kGridThreads) {
        FinalKp f = fin[k];
        int level;
        const int cell = desc_cell_of(plan, f, level);
        if (cell < 0) continue;
        f.cand = (uint32_t)k;   // vector index: the order the chains follow, and the output slot
        pl[atomicAdd(&cnt[cell], 1)] = f;
    }
}

// minimum of a non-negative int over the wave, in a scalar register: butterfly inside each row of 16 lanes (DPP), then the
// four rows by readlane
__device__ __forceinline__ int wave_min_nonneg(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x140, 0xf, 0xf, false));   // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

template <int Y>
__device__ __forceinline__ float quad_bcast_f(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), Y * 0x55, 0xF, 0xF, true));
}
template <int Y>
__device__ __forceinline__ unsigned quad_bcast_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, Y * 0x55, 0xF, 0xF, true);
}
// value of the previous lane of the quad (lane 0 receives lane 3's: unused)
__device__ __forceinline__ float quad_prev_f(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x93 /* quad_perm [3,0,1,2] */, 0xF, 0xF, true));
}

// The 16 cell histograms of one keypoint (alg::orientationHistogram8, algorithms.cpp:135-150), shared by the descriptor kernels
// below.  A DPP quad holds one 4x4 cell: lane y of the quad has the cell's row y, val[i] / bin[i] = its sample at x = i; lane l
// of the quad accumulates the bins b0 = 2l and b1 = 2l + 1 over the 16 samples in the reference's order (x outer, y inner).
// Written out in assembly: the four bins of a column (one per lane of the quad) are packed into one dword (byte y = bin of row
// y: a shift and two DPP ORs per column), so that a sample costs two SDWA byte compares, two DPP additions (val of row y + h)
// and two selects - six instructions, none of them a move; the compiler's form took nine (its packed additions keep the DPP
// broadcasts as separate moves, with wait states in front of them).
#define SIFT_HIST_SAMPLE(Y)                                                                                  \
    "v_cmp_eq_u32_sdwa vcc, %[p], %[b0] src0_sel:BYTE_" #Y " src1_sel:DWORD\n\t"                              \
    "v_add_f32_dpp %[t], %[v], %[h0] quad_perm:[" #Y "," #Y "," #Y "," #Y "] row_mask:0xf bank_mask:0xf\n\t"  \
    "v_cndmask_b32 %[h0], %[h0], %[t], vcc\n\t"                                                               \
    "v_cmp_eq_u32_sdwa vcc, %[p], %[b1] src0_sel:BYTE_" #Y " src1_sel:DWORD\n\t"                              \
    "v_add_f32_dpp %[t], %[v], %[h1] quad_perm:[" #Y "," #Y "," #Y "," #Y "] row_mask:0xf bank_mask:0xf\n\t"  \
    "v_cndmask_b32 %[h1], %[h1], %[t], vcc\n\t"
__device__ __forceinline__ void cell_histograms(const float (&val)[4], const unsigned (&bin)[4], int lane, unsigned b0, unsigned b1,
                                                float& h0, float& h1) {
    const unsigned sh = 8u * (unsigned)(lane & 3);
    unsigned pk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pk[i] = bin[i] << sh;
    // byte y of pk[i] = bin of (column i, row y), in every lane of the quad.  (A DPP operand must not have been written by
    // the two instructions before: the four columns are interleaved, with explicit wait states at the block's start.)
    asm volatile(
        "s_nop 1\n\t"
        "v_or_b32_dpp %[p0], %[p0], %[p0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_or_b32_dpp %[p1], %[p1], %[p1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_or_b32_dpp %[p2], %[p2], %[p2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_or_b32_dpp %[p3], %[p3], %[p3] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_or_b32_dpp %[p0], %[p0], %[p0] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_or_b32_dpp %[p1], %[p1], %[p1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_or_b32_dpp %[p2], %[p2], %[p2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_or_b32_dpp %[p3], %[p3], %[p3] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        : [p0] "+v"(pk[0]), [p1] "+v"(pk[1]), [p2] "+v"(pk[2]), [p3] "+v"(pk[3]));
    float t;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        asm volatile(
            "s_nop 1\n\t"
            SIFT_HIST_SAMPLE(0) SIFT_HIST_SAMPLE(1) SIFT_HIST_SAMPLE(2) SIFT_HIST_SAMPLE(3)
            : [h0] "+v"(h0), [h1] "+v"(h1), [t] "=&v"(t)
            : [p] "v"(pk[i]), [v] "v"(val[i]), [b0] "v"(b0), [b1] "v"(b1)
            : "vcc");
}
#undef SIFT_HIST_SAMPLE

__global__ __launch_bounds__(256, 8) void descriptor_wave_kernel(const DevPlan* __restrict__ plan, DescGridLevel lv,
                                                                 const int* __restrict__ cell_off,
                                                                 const FinalKp* __restrict__ pool, int pool_cap,
                                                                 const long long* __restrict__ out_base,
                                                                 sift_hip_keypoint* __restrict__ kp_out,
                                                                 float* __restrict__ desc_out, long long out_cap, int n_images,
                                                                 int chunks, int dbg_arg) {
    const int dbg = dbg_arg & kDiagMask;   // measurement build only (common.h)
    __shared__ __attribute__((aligned(8))) W16Entry s_w16t[kW16Size];
    __shared__ __attribute__((aligned(16))) uint2 s_list[4][64];   // per wave: preceding neighbours in vector order (orientation bits, table offset)
    __shared__ __attribute__((aligned(16))) unsigned s_keys[4][68];   // per wave: their vector indices, compacted (+ sentinels)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = lv.w, h = lv.h, D = lv.dogs;
    const int ly = lane & 15, lx0 = (lane >> 4) * 4;   // this lane's row and first column inside a window
    const int tbase = kW16Bias + lx0 * kW16Stride + ly;
    const unsigned b0 = 2u * (unsigned)(lane & 3), b1 = b0 + 1u;   // the two bins this lane accumulates for its cell
    for (int i = tid; i < kW16Size; i += 256) s_w16t[i] = W16Entry{0.0f, 0u};
    // Work units = (image, chunk of the image's records): workgroups with equal blockIdx % 8 (observed to share an XCD and
    // its 4 MB L2; a different placement only costs speed) take the units u = blockIdx % 8, + 8, + 16 ... one after the
    // other, all of them striding through the SAME unit at a time.  The records of a unit are consecutive grid cells, i.e. a
    // band of the image a few cells high, so the windows the XCD's waves read at any moment overlap in its L2 instead of
    // every L2 holding a slice of every image of the batch.
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3, nslots = (int)gridDim.x >> 3;
    for (int unit = xcd; unit < n_images * chunks; unit += 8) {
    const int img = unit / chunks, chunk = unit - img * chunks;
    const int* __restrict__ coff = cell_off + (size_t)img * (size_t)(lv.cells_per_image + 1) + lv.cell_base;
    const int r_begin = coff[0], r_count = coff[lv.cw * lv.ch] - r_begin;
    const int e_begin = r_begin + (int)((long long)r_count * chunk / chunks), e_end = r_begin + (int)((long long)r_count * (chunk + 1) / chunks);
    if (e_begin + slot * 4 >= e_end) continue;   // nothing for this workgroup in this unit
    // weighting(x, y) (sift.cpp:87-90), x-major: index bias + 20 * x + y.  Entries outside 0..15 x 0..15 are only ever
    // read by lanes that then discard them.
    __syncthreads();   // the previous unit's readers are done
    s_w16t[kW16Bias + (tid & 15) * kW16Stride + (tid >> 4)] = W16Entry{lv.w16[(size_t)img * 256 + tid], 0xffffffffu};
    __syncthreads();
    float wself[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wself[i] = s_w16t[tbase + i * kW16Stride].w;
    const size_t img_off = (size_t)img * (size_t)w * (size_t)h;
    const float* __restrict__ gm = lv.mag + img_off;
    const float* __restrict__ go = lv.ori + img_off;
    const float* __restrict__ gg = lv.gauss + img_off;
    const FinalKp* __restrict__ pl = pool + (size_t)img * (size_t)pool_cap;
    const long long obase = out_base[img];

    // Scalar prefetch over the wave's keypoints: a keypoint's record (x, y, orientation, vector index) is fetched two
    // iterations ahead and the bounds of its three cell-row runs of records one iteration ahead, so an iteration starts with
    // everything its vector loads (window pixels of the three maps, one neighbour record per lane) need and never waits for a
    // chain of dependent loads.  (Fetching the window pixels an iteration ahead as well was measured: no gain, the extra
    // registers spill.)
    const int estride = nslots * 4;
    const int e0 = e_begin + slot * 4 + wave;
    struct Rec { int k; float theta; unsigned xy, oi; };
    struct Ranges { int s0, s1, s2, n0, n1, n2; };
    struct Window { f4u o, m, g; };
    auto load_rec = [&](int e) {
        const uint4 v = *reinterpret_cast<const uint4*>(&pl[min(e, e_end - 1)]);   // past the end: a harmless repeat
        Rec r;
        r.k = (int)v.x; r.theta = __uint_as_float(v.y); r.xy = v.z; r.oi = v.w;
        return r;
    };
    auto load_ranges = [&](unsigned xy) {   // the 3x3 cells around (x, y): three runs of consecutive records, one per cell row
        const int ccx = (int)(xy & 0xffffu) >> kCellShift, ccy = (int)(xy >> 16) >> kCellShift;
        const int c0 = max(ccx - 1, 0), c1 = min(ccx + 1, lv.cw - 1);
        const int y0 = max(ccy - 1, 0), y2 = min(ccy + 1, lv.ch - 1);
        // unconditional loads (rows clamped into the grid), the counts of rows outside it zeroed afterwards
        const int a0 = coff[y0 * lv.cw + c0], b0_ = coff[y0 * lv.cw + c1 + 1];
        const int a1 = coff[ccy * lv.cw + c0], b1_ = coff[ccy * lv.cw + c1 + 1];
        const int a2 = coff[y2 * lv.cw + c0], b2_ = coff[y2 * lv.cw + c1 + 1];
        Ranges g;
        g.s0 = a0; g.n0 = ccy - 1 >= 0 ? b0_ - a0 : 0;
        g.s1 = a1; g.n1 = b1_ - a1;
        g.s2 = a2; g.n2 = ccy + 1 < lv.ch ? b2_ - a2 : 0;
        return g;
    };
    auto uniform_rec = [](const Rec& r) {   // the record is the same in every lane: keep it in scalar registers
        Rec u;
        u.k = __builtin_amdgcn_readfirstlane(r.k);
        u.theta = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(r.theta)));
        u.xy = (unsigned)__builtin_amdgcn_readfirstlane((int)r.xy);
        u.oi = (unsigned)__builtin_amdgcn_readfirstlane((int)r.oi);
        return u;
    };
    // (every keypoint of the grid passed the bounds test, so its window lies inside the image)
    auto load_window = [&](unsigned xy) {
        size_t o = (size_t)((int)(xy >> 16) - kRegion + ly) * (size_t)w + (size_t)((int)(xy & 0xffffu) - kRegion + lx0);
        if (dbg & 8) o &= ~(size_t)3;   // timing only: 16-byte aligned window loads (wrong pixels)
        if (dbg & 128) o = (size_t)ly * (size_t)w + (size_t)lx0 + (size_t)((xy >> 4) & 0xfffu) * 16u;   // timing only: every window from the image's first rows (cache hits)
        Window wd;
        wd.o = (f4u)(1.0f); wd.m = (f4u)(2.0f); wd.g = (f4u)(3.0f);
        if (!(dbg & 4)) {               // timing only: no window loads
            wd.o = *reinterpret_cast<const f4u*>(go + o);
            wd.m = *reinterpret_cast<const f4u*>(gm + o);
            wd.g = *reinterpret_cast<const f4u*>(gg + o);
        }
        return wd;
    };
    auto entry_of = [](const Ranges& g, int i) {   // i-th record of the three runs
        return i < g.n0 ? g.s0 + i : (i < g.n0 + g.n1 ? g.s1 + (i - g.n0) : g.s2 + (i - g.n0 - g.n1));
    };
    auto load_cand = [&](const Ranges& g) {   // one neighbour record per lane (crowded neighbourhoods are rescanned instead)
        const int T = g.n0 + g.n1 + g.n2;
        uint4 c = make_uint4(0u, 0u, 0u, 0u);
        if (lane < T && T <= 64) c = *reinterpret_cast<const uint4*>(&pl[entry_of(g, lane)]);
        return c;
    };
    Rec u0 = uniform_rec(load_rec(e0));
    Rec r1 = load_rec(e0 + estride);
    Ranges g0 = load_ranges(u0.xy);

    for (int e = e0; e < e_end; e += estride) {
        // the neighbour records (one per lane) travel together with the window's pixels
        const uint4 c0v = load_cand(g0);
        const Window w0 = load_window(u0.xy);
        // ---- prefetch stages --------------------------------------------------------------------------------------
        const Rec u1 = uniform_rec(r1);             // arrived during the previous iteration
        r1 = load_rec(e + 2 * estride);
        const Ranges g1 = load_ranges(u1.xy);
        // ---- this wave's keypoint (wave-uniform: scalar registers) ----------------------------------------------
        const int myk = u0.k;
        const float mytheta = u0.theta;
        const unsigned myoi = u0.oi;   // octave | index << 16
        const int px = (int)(u0.xy & 0xffffu), py = (int)(u0.xy >> 16);
        const int T = g0.n0 + g0.n1 + g0.n2;
        const uint4 c = c0v;
        float vo[4] = {w0.o.x, w0.o.y, w0.o.z, w0.o.w}, vm[4] = {w0.m.x, w0.m.y, w0.m.z, w0.m.w};
        const float vg[4] = {w0.g.x, w0.g.y, w0.g.z, w0.g.w};

        // one preceding neighbour q: the lane's pixels that q's window covers receive q's additions (sift.cpp:80-92).
        // The four table entries are fetched by `weights_at` (which can run one neighbour ahead) and consumed by `apply_w`.
        struct Wq { W16Entry v[4]; };
        // dx, dy: p's window-local (x, y) is q's (x + dx, y + dy); toff = dx * kW16Stride + dy
        auto weights_at = [&](int toff) {
            Wq r;
#pragma unroll
            for (int i = 0; i < 4; ++i) r.v[i] = s_w16t[tbase + toff + i * kW16Stride];
            return r;
        };
        auto apply_w = [&](unsigned qtheta_bits, const Wq& wq) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vo[i] = vo[i] + __uint_as_float(qtheta_bits & wq.v[i].mask);   // + theta inside q's window, + 0.0f outside
                vm[i] = vm[i] + wq.v[i].w;                                     // + 0.0f from the table's padding outside
            }
        };
        auto apply = [&](unsigned qxy, float qtheta) {
            const int dx = px - (int)(qxy & 0xffffu), dy = py - (int)(qxy >> 16);
            apply_w(__float_as_uint(qtheta), weights_at(dx * kW16Stride + dy));
        };

        // ---- neighbours: the 3x3 cells around p's --------------------------------------------------------------------
        if (!(dbg & 1)) {
            auto precedes = [&](const uint4& c) {   // earlier in the vector and close enough to share a pixel
                const int qx = (int)(c.z & 0xffffu), qy = (int)(c.z >> 16);
                return (int)c.x < myk && (unsigned)(qx - px + 15) < 31u && (unsigned)(qy - py + 15) < 31u;
            };
            if (T <= 64) {
                // One record per lane.  The preceding neighbours are ranked by vector index: their indices are compacted into
                // this wave's LDS key list (position = number of passing lanes below), and every lane counts the keys smaller
                // than its own with broadcast 16-byte reads — a compare and an add per key instead of a trip of a scalar loop
                // (find the next set bit, readlane, compare, add) per neighbour.  The neighbours are then dropped into the LDS
                // list at their rank and simply walked.  (Measured alternatives, no faster: a wave minimum per neighbour;
                // finding the lane of rank r by ballot with the weighting values fetched one neighbour ahead.)
                const bool pass = lane < T && precedes(c);
                const unsigned long long todo = __ballot(pass);
                const int n_prev = __popcll(todo);
                if (n_prev) {
                    unsigned* keys = s_keys[wave];
                    const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(todo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)todo, 0u));
                    if (pass) keys[pos] = c.x;
                    if (lane < 4) keys[n_prev + lane] = 0xffffffffu;   // the last 16-byte read may run past the keys: never smaller
                    __builtin_amdgcn_wave_barrier();
                    int rank = 0;
                    for (int ch = 0; ch < n_prev; ch += 4) {
                        const uint4 k4 = *reinterpret_cast<const uint4*>(keys + ch);
                        rank += (k4.x < c.x ? 1 : 0) + (k4.y < c.x ? 1 : 0) + (k4.z < c.x ? 1 : 0) + (k4.w < c.x ? 1 : 0);
                    }
                    // each lane works out what the walk needs of ITS neighbour once (offsets, table address), so the walk's body
                    // is a broadcast read and the additions, two neighbours per trip
                    uint2* list = s_list[wave];
                    if (pass) {
                        const int dx = px - (int)(c.z & 0xffffu), dy = py - (int)(c.z >> 16);
                        list[rank] = make_uint2(c.y, (unsigned)(dx * kW16Stride + dy));
                    }
                    __builtin_amdgcn_wave_barrier();
                    int r = 0;
                    for (; r + 1 < n_prev; r += 2) {
                        const uint4 q01 = *reinterpret_cast<const uint4*>(&list[r]);   // two neighbours per 16-byte broadcast read
                        const Wq wq0 = weights_at((int)q01.y), wq1 = weights_at((int)q01.w);
                        apply_w(q01.x, wq0);
                        apply_w(q01.z, wq1);
                    }
                    if (r < n_prev) {
                        const uint2 q0 = list[r];
                        apply_w(q0.x, weights_at((int)q0.y));
                    }
                    __builtin_amdgcn_wave_barrier();   // the list is rewritten for the wave's next keypoint
                }
            } else {
                // crowded neighbourhood: take the preceding neighbours in order by rescanning the records
                int last = -1;
                for (;;) {
                    int bk = 0x7fffffff;
                    unsigned bxy = 0u, bth = 0u;
                    for (int base = 0; base < T; base += 64) {
                        const int i = base + lane;
                        if (i < T) {
                            const uint4 c = *reinterpret_cast<const uint4*>(&pl[entry_of(g0, i)]);
                            if (precedes(c) && (int)c.x > last && (int)c.x < bk) { bk = (int)c.x; bxy = c.z; bth = c.y; }
                        }
                    }
                    const int m = wave_min_nonneg(bk);
                    if (m == 0x7fffffff) break;
                    const int src = (int)__builtin_ctzll(__ballot(bk == m));
                    apply((unsigned)__builtin_amdgcn_readlane((int)bxy, src), __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)bth, src)));
                    last = m;
                }
            }
        }
        // ---- p's own update (sift.cpp:80-92) and the histogram inputs (algorithms.cpp:135-150) ---------------------------
        float val[4];
        unsigned bin[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float no = vo[i] + mytheta;
            const float nm = vm[i] + wself[i];
            val[i] = nm * vg[i];
            bin[i] = hist8_bin(no);
        }
        // ---- 16 cell histograms: a quad holds one cell; samples in x-outer / y-inner order ------------------------------
        float h0 = 0.0f, h1 = 0.0f;
        if (!(dbg & 2)) {
            cell_histograms(val, bin, lane, b0, b1, h0, h1);
            // alg::normalizeVector (algorithms.cpp:210-223): length = ((0 + b0) + b1) + ... + b7, handed along the quad
            float run = 0.0f;
            run += h0; run += h1;                       // lane 0 of the quad: bins 0, 1
            float acc = run;
#pragma unroll
            for (int step = 1; step < 4; ++step) {      // lane `step` continues from lane step-1's sum
                const float prev = quad_prev_f(acc);
                float t = prev;
                t += h0; t += h1;
                acc = ((lane & 3) == step) ? t : acc;
            }
            const float length = quad_bcast_f<3>(acc);
            if (!(length == 0.0f)) {
                h0 = h0 / length;
                h1 = h1 / length;
            }
        }
        const long long ok = obase + (long long)myk;
        if (ok < out_cap && !((dbg & 16) && h0 == 12345.0f)) {   // (the output arrays are sized before the final counts reach the host: see run_batch)
            typedef float f2nt __attribute__((ext_vector_type(2)));
            __builtin_nontemporal_store((f2nt){h0, h1}, reinterpret_cast<f2nt*>(desc_out + (size_t)ok * 128 + (size_t)(2 * lane)));
            if (lv.wire_sums != nullptr) {
                // the counting pass of the sparse wire format (kernels_wire.hip: wire_count_kernel) while the 128 floats are
                // still in registers: floats that are not +0.0f outside bin 7, per block of 64 output slots
                const bool b0 = __float_as_uint(h0) != 0u, odd7 = (lane & 3) == 3;
                const bool b1 = !odd7 && __float_as_uint(h1) != 0u;
                const int nset = __popcll(__ballot(b0)) + __popcll(__ballot(b1));
                const unsigned long long bin7 = __ballot(odd7 && __float_as_uint(h1) != 0u);
                if (lane == 0) {
                    atomicAdd(&lv.wire_sums[1 + (ok >> 6)], nset);
                    if (bin7 != 0ull) atomicOr(&lv.wire_sums[0], 1);
                }
            }
            if (lane == 0) {
                sift_hip_keypoint r;
                r.scale = plan->dog_scale[(myoi & 0xffffu) * (unsigned)D + (myoi >> 16)];
                r.orientation = mytheta;
                r.x = (uint16_t)px; r.y = (uint16_t)py;
                r.octave = (uint16_t)(myoi & 0xffffu); r.index = (uint16_t)(myoi >> 16);
                r.filtered = 0; r.has_descriptor = 1; r.reserved = 0;
                kp_out[ok] = r;
            }
        }
        u0 = u1;
        g0 = g1;
    }
    }   // units
}

void launch_w16(hipStream_t s, const DevPlan& plan, int level, const float* d_taps16, int radius16) {
    const int oct = level / (plan.dogs + 1);
    hipLaunchKernelGGL(w16_kernel, dim3((unsigned)plan.n_images), dim3(256), 0, s,
                       (const float*)plan.gauss[level], plan.w16[level], plan.w[oct], plan.h[oct], d_taps16,
                       radius16);
}

// grid of 16 px cells over the final keypoints of every image
void launch_desc_grid(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan, const FinalKp* d_final, const int* d_final_cnt,
                      int final_cap, int* d_cell_cnt, int* d_cell_off, FinalKp* d_pool, int pool_cap, long long* d_out_base,
                      bool compute_base, sift_hip_keypoint* d_kp_out, float* d_desc_out, long long out_cap) {
    if (plan.desc_cells_per_image <= kGridLdsCells)
        hipLaunchKernelGGL(desc_grid_kernel<true>, dim3((unsigned)plan.n_images), dim3(kGridThreads), 0, s, d_plan, d_final, d_final_cnt,
                           final_cap, d_cell_cnt, d_cell_off, d_pool, pool_cap, d_out_base, compute_base ? 1 : 0, d_kp_out, d_desc_out, out_cap);
    else
        hipLaunchKernelGGL(desc_grid_kernel<false>, dim3((unsigned)plan.n_images), dim3(kGridThreads), 0, s, d_plan, d_final, d_final_cnt,
                           final_cap, d_cell_cnt, d_cell_off, d_pool, pool_cap, d_out_base, compute_base ? 1 : 0, d_kp_out, d_desc_out, out_cap);
}

void launch_descriptors_wave(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan, int level, const int* d_cell_off,
                             const FinalKp* d_pool, int pool_cap, const long long* d_out_base, sift_hip_keypoint* d_kp_out,
                             float* d_desc_out, long long out_cap, int dbg, int* d_wire_sums, hipEvent_t ev_start, hipEvent_t ev_stop) {
    DescGridLevel lv;
    const int oct = level / (plan.dogs + 1);
    lv.w = plan.w[oct]; lv.h = plan.h[oct]; lv.dogs = plan.dogs;
    lv.cw = plan.desc_cw[level]; lv.ch = plan.desc_ch[level]; lv.cell_base = plan.desc_cell_base[level];
    lv.cells_per_image = plan.desc_cells_per_image;
    lv.mag = plan.mag[level]; lv.ori = plan.ori[level]; lv.gauss = plan.gauss[level]; lv.w16 = plan.w16[level];
    lv.wire_sums = d_wire_sums;
    // 2048 workgroups of four waves (8 per CU); whole images per XCD when there are at least 8, eighths of an image otherwise
    const int chunks = plan.n_images >= 8 ? 1 : 8;
    const unsigned nwg = (dbg & kDiagMask & 64) ? 512u : ((dbg & kDiagMask & 32) ? 1024u : 2048u);   // timing only (measurement build): fewer resident waves
    hipExtLaunchKernelGGL(descriptor_wave_kernel, dim3(nwg), dim3(256), 0, s, ev_start, ev_stop, 0, d_plan, lv, d_cell_off, d_pool, pool_cap, d_out_base,
                       d_kp_out, d_desc_out, out_cap, plan.n_images, chunks, dbg);
}

// The runtime builds a translation unit's device code on the first launch of any of its kernels, and two host threads that make
// their first launches at the same time (several contexts, one thread each) were seen to crash inside that step
// (tools/asan_example.sh: SEGV below hipLaunchKernel).  sift_hip_create touches every unit once, under a lock.
__global__ void tu_probe_desc_kernel() {}
void tu_touch_desc(hipStream_t s) { hipLaunchKernelGGL(tu_probe_desc_kernel, dim3(1), dim3(1), 0, s); }

}  // namespace sift_hip
