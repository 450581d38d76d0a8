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
#include <cstdlib>

#include <hip/hip_ext.h>

#include "common.h"
#include "lds_tile.h"

#pragma clang fp contract(off)

namespace sift_hip {

__device__ __forceinline__ int reflect_clamp(int p, int n) {
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * (n - 1) - p : p;
    // lanes that only feed outputs outside the image may still be out of range: keep them legal
    return p < 0 ? 0 : (p >= n ? n - 1 : p);
}

// ---------------------------------------------------------------------------------------------
// Fused blur, tile form (radii above 14, small levels, small batches; the streaming form further down takes the
// rest).  A persistent workgroup walks 64x64 output tiles:
//   1. the (64+2*RA) x (64+2R) source tile (RA = R rounded up to 4, halo reflected at the image
//      border) is fetched with row-coalesced 16-byte HBM loads into REGISTERS one tile ahead, so the
//      HBM latency of tile t+1 hides under the arithmetic of tile t, then written to LDS;
//   2. row pass LDS -> registers (4 consecutive outputs per thread from ds_read_b128 windows) -> LDS;
//   3. column pass LDS -> registers (4 columns x 4 rows per thread, conflict-free ds_read_b128);
//      DoG = 128 + (blurred - source) from the source tile still in LDS;
//   4. 16-byte coalesced stores.
// The intermediate never touches HBM: 4 B read + 4 B (8 B with DoG) written per pixel.
// Barriers are LDS-only (s_waitcnt lgkmcnt(0); s_barrier) so the prefetch stays in flight.
// Taps are read with wave-uniform constant indices => scalar loads into SGPRs.
// Workgroup ids are dealt round-robin over the 8 XCDs, so each XCD (= ids equal mod 8) is given a
// contiguous run of tiles: neighbouring tiles' halos then hit that XCD's L2.
// Tiles that touch the image border, or images whose rows are not 16-byte aligned, take a scalar
// load / store path with per-element reflection.
// ---------------------------------------------------------------------------------------------
// gcd_ce, lds_read_window<N>, lds_barrier: lds_tile.h

template <int R, bool DOG, int TH = 64>
__global__ __launch_bounds__(256, (R <= 8 ? 3 : ((R <= 24 || (TH <= 48 && R <= 28)) ? 2 : 1))) void blur_fused_kernel(const float* __restrict__ in,
                                                         float* __restrict__ out,
                                                         float* __restrict__ dog, int w, int h,
                                                         int tiles_x, int tiles_y, int total_tiles,
                                                         int vec_ok, const float* __restrict__ taps) {
    constexpr int TW = 64;
    constexpr int RA = (R + 3) & ~3;
    constexpr int PAD = RA - R;
    constexpr int SWA = TW + 2 * RA;       // LDS row length, multiple of 4
    constexpr int SH = TH + 2 * R;
    constexpr int ROW4 = SWA / 4;          // float4 per source row
    constexpr int NL4 = SH * ROW4;         // float4 per source tile
    constexpr int NLD = (NL4 + 255) / 256; // prefetch registers (float4) per thread
    constexpr int NT = 2 * R + 1;
    __shared__ __attribute__((aligned(16))) float s_src[SH * SWA];
    __shared__ __attribute__((aligned(16))) float s_mid[SH * TW];
    float4* s_src4 = reinterpret_cast<float4*>(s_src);
    float4* s_mid4 = reinterpret_cast<float4*>(s_mid);

    const int tid = threadIdx.x;
    // tile schedule: XCD x (= id mod 8) owns tiles [x*chunk, (x+1)*chunk)
    int t, t_end, t_step;
    if ((gridDim.x & 7u) == 0) {
        const int chunk = (total_tiles + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        t = xcd * chunk + (int)(blockIdx.x >> 3);
        t_end = min(xcd * chunk + chunk, total_tiles);
        t_step = (int)(gridDim.x >> 3);
    } else {
        t = blockIdx.x;
        t_end = total_tiles;
        t_step = gridDim.x;
    }
    const int tiles = tiles_x * tiles_y;

    float4 pre[NLD];
    auto load_tile = [&](int tile) {
        const int img = tile / tiles;
        const int t2 = tile - img * tiles;
        const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
        const int x0 = tx * TW, y0 = ty * TH;
        const float* __restrict__ src = in + (size_t)img * (size_t)w * (size_t)h;
        const bool interior = vec_ok && x0 - RA >= 0 && x0 + TW + RA <= w && y0 - R >= 0 && y0 + TH + R <= h;
        if (interior) {
            const float* __restrict__ base = src + (size_t)(y0 - R) * (size_t)w + (size_t)(x0 - RA);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = tid + 256 * i;
                if (e < NL4) {
                    const int ly = e / ROW4, c4 = e - ly * ROW4;
                    pre[i] = *reinterpret_cast<const float4*>(base + (size_t)ly * (size_t)w + (size_t)(4 * c4));
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = tid + 256 * i;
                if (e < NL4) {
                    const int ly = e / ROW4, c4 = e - ly * ROW4;
                    const size_t row = (size_t)reflect_clamp(y0 - R + ly, h) * (size_t)w;
                    const int gx = x0 - RA + 4 * c4;
                    if (vec_ok && gx >= 0 && gx + 3 < w) {
                        // only the row is reflected: the four columns lie inside the image (gx is a multiple of 4)
                        pre[i] = *reinterpret_cast<const float4*>(src + row + (size_t)gx);
                    } else {
                        pre[i].x = src[row + (size_t)reflect_clamp(gx + 0, w)];
                        pre[i].y = src[row + (size_t)reflect_clamp(gx + 1, w)];
                        pre[i].z = src[row + (size_t)reflect_clamp(gx + 2, w)];
                        pre[i].w = src[row + (size_t)reflect_clamp(gx + 3, w)];
                    }
                }
            }
        }
    };

    if (t < t_end) load_tile(t);
    while (t < t_end) {
        // 1. prefetched source tile -> LDS
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + 256 * i;
            if (e < NL4) s_src4[e] = pre[i];
        }
        lds_barrier();
        const int tn = t + t_step;
        if (tn < t_end) load_tile(tn);  // stays in flight through both passes

        // 2. row pass.  A wave reads 4 rows x 16 float4 per instruction; the rows are taken RG apart so
        // that their LDS offsets agree modulo 256 B, the row spacing the ds_read_b128 lane groups are
        // conflict-free for.
        constexpr int RG = 16 / gcd_ce(ROW4, 16);
        constexpr int NBLK = (SH + 4 * RG - 1) / (4 * RG);      // blocks of 4*RG rows
        // the bottom tile row of an image needs only the source rows its (fewer) output rows reach
        const int tile_y0 = ((t % tiles) / tiles_x) * TH;
        const int rows_used = min(SH, h - tile_y0 + 2 * R);
#pragma unroll 1
        for (int wi = tid >> 6; wi < NBLK * RG; wi += 4) {
            const int ly = (wi / RG) * (4 * RG) + (wi % RG) + RG * ((tid >> 4) & 3);
            const int q = tid & 15;
            if (ly >= rows_used) continue;
            constexpr int NV = PAD + 4 + 2 * R;
            constexpr int NV4 = (NV + 3) / 4;
            float v[NV4 * 4];
            const float4* p4 = &s_src4[ly * ROW4 + q];
            float4 f4[NV4];
            lds_read_window<NV4>(p4, f4);
#pragma unroll
            for (int c = 0; c < NV4; ++c) {
                const float4 f = f4[c];
                v[4 * c + 0] = f.x;
                v[4 * c + 1] = f.y;
                v[4 * c + 2] = f.z;
                v[4 * c + 3] = f.w;
            }
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const float tap = taps[NT - 1 - k];
                a0 += tap * v[PAD + k];
                a1 += tap * v[PAD + k + 1];
                a2 += tap * v[PAD + k + 2];
                a3 += tap * v[PAD + k + 3];
            }
            s_mid4[ly * (TW / 4) + q] = make_float4(a0, a1, a2, a3);
        }
        lds_barrier();

        // 3. column pass: 4 columns x PY rows per thread
        constexpr int PY = TH / 16;
        const int cg = tid & 15, rg = tid >> 4;
        float4 acc[PY];
#pragma unroll
        for (int i = 0; i < PY; ++i) acc[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (tile_y0 + rg * PY < h)   // rows below the image: nothing to compute or store
#pragma unroll
        for (int k = 0; k < PY + 2 * R; ++k) {
            const float4 m = s_mid4[(rg * PY + k) * (TW / 4) + cg];
#pragma unroll
            for (int i = 0; i < PY; ++i) {
                if (k - i >= 0 && k - i <= 2 * R) {
                    const float tap = taps[NT - 1 - (k - i)];
                    acc[i].x += tap * m.x;
                    acc[i].y += tap * m.y;
                    acc[i].z += tap * m.z;
                    acc[i].w += tap * m.w;
                }
            }
            // keep the unrolled loop from hoisting every LDS read to the top (VGPR blow-up)
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        // 4. stores
        {
            const int img = t / tiles;
            const int t2 = t - img * tiles;
            const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
            const int x = tx * TW + 4 * cg;
            const size_t img_off = (size_t)img * (size_t)w * (size_t)h;
#pragma unroll
            for (int i = 0; i < PY; ++i) {
                const int y = ty * TH + rg * PY + i;
                if (y < h && x < w) {
                    const size_t o = img_off + (size_t)y * (size_t)w + (size_t)x;
                    float4 d4;
                    if (DOG) {
                        const float4 prev = s_src4[(R + rg * PY + i) * ROW4 + (RA / 4) + cg];
                        const float dx = acc[i].x - prev.x, dy = acc[i].y - prev.y;
                        const float dz = acc[i].z - prev.z, dw = acc[i].w - prev.w;
                        d4 = make_float4(128.0f + dx, 128.0f + dy, 128.0f + dz, 128.0f + dw);
                    }
                    if (vec_ok && x + 3 < w) {
                        if (out) *reinterpret_cast<float4*>(out + o) = acc[i];   // out == nullptr: only the DoG is wanted
                        if (DOG) *reinterpret_cast<float4*>(dog + o) = d4;
                    } else {
                        const float av[4] = {acc[i].x, acc[i].y, acc[i].z, acc[i].w};
                        const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (x + j < w) {
                                if (out) out[o + j] = av[j];
                                if (DOG) dog[o + j] = dv[j];
                            }
                    }
                }
            }
        }
        lds_barrier();  // s_src / s_mid are rewritten by the next tile
        t = tn;
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming blur (small radii, rows that are 16-byte aligned).  Every WAVE is on its own: it owns a
// strip of up to 256 columns (4 per lane) and a chunk of rows, and walks the chunk top to bottom:
//   * the source row (plus RA reflected halo columns each side) is fetched PF rows ahead into
//     registers, then dropped into a per-wave LDS row (a ring of R+1 rows when the DoG needs the
//     source again R rows later);
//   * row pass: ds_read_b128 window -> 4 consecutive outputs per lane, the reference's order;
//   * column pass in REGISTERS: the 2R+1 partial sums of the lane's 4 columns slide by one each row,
//         A[j] = A[j+1] + tap[j] * mid      (A[2R] = 0 + tap[2R] * mid)
//     so output row y receives its terms for source rows y-R .. y+R in ascending order from 0.0f,
//     exactly the reference's sequence; A[0] is complete after the step and is stored.
// No workgroup barrier, no intermediate tile in LDS, no vertical halo inside a chunk: HBM sees each
// source row once per chunk (+2R rows of run-in) and LDS traffic is the row windows only.
// ---------------------------------------------------------------------------------------------
constexpr int kStreamPF = 4;  // source rows in flight per wave (registers)
// run-in rows before the first output: 2R rounded up to whole unrolled bodies (the extra leading rows
// only feed partial sums that are never stored)
constexpr int stream_runin(int r) { return (2 * r + kStreamPF - 1) / kStreamPF * kStreamPF; }
// waves per SIMD the register budget is cut for (512 VGPRs per lane per SIMD)
constexpr int stream_occ(int r, int cpl) { return cpl == 4 ? (r <= 8 ? 3 : 2) : (r <= 8 ? 4 : r <= 12 ? 3 : 2); }

// decimation of the blurred image on the way out (alg::reduceToNextLevel, algorithms.cpp:24-36): only the pixels the
// nearest-neighbour resampling keeps are stored, straight into the next octave's first level
struct StreamDecimate {
    const int* inv_x;   // source column -> destination column, or -1
    const int* inv_y;   // source row -> destination row, or -1
    int wd, hd;         // destination size
    float* dump;        // >= 64 * CPL floats nobody reads: where the unselected pixels go (no branch around a store)
};

template <int R, bool DOG, int CPL, bool DEC = false>
__global__ __launch_bounds__(256, stream_occ(R, CPL)) void blur_stream_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          float* __restrict__ dog, int w, int h, int strips,
                                                          int strip_w, int chunks, int chunk_h, int total_units,
                                                          const float* __restrict__ taps, StreamDecimate dec) {
    static_assert(!(DEC && DOG), "the decimating variant has no DoG output");
    constexpr int PF = kStreamPF;
    constexpr int RI = stream_runin(R);
    constexpr int E = RI - 2 * R;
    constexpr int RA = (R + CPL - 1) / CPL * CPL;  // halo columns each side, a whole number of lane vectors
    constexpr int PAD = RA - R;
    constexpr int NT = 2 * R + 1;
    constexpr int ROWF = 64 * CPL + 2 * RA;  // floats per LDS row
    constexpr int DP = DOG ? R + 2 : 1;  // ring depth: rows s-R .. s+1 are live when the DoG reads its source
    constexpr int NV = PAD + CPL + 2 * R;
    constexpr int NV4 = (NV + CPL - 1) / CPL;  // lane vectors per window
    __shared__ __attribute__((aligned(16))) float s_ring[4][DP * ROWF];
    typedef float f4v __attribute__((ext_vector_type(CPL)));  // CPL consecutive columns of one row

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int unit = (int)blockIdx.x * 4 + wave;
    if (unit >= total_units) return;
    const int per_img = strips * chunks;
    const int img = unit / per_img;
    const int rem = unit - img * per_img;
    const int chunk = rem / strips;
    const int strip = rem - chunk * strips;
    const int xs = strip * strip_w;
    const int sw = min(strip_w, w - xs);
    // every chunk is chunk_h rows (a multiple of PF); the last one is pulled up to end at the image's
    // last row and rewrites a few rows of its neighbour with the same values
    const int y0 = min(chunk * chunk_h, h - chunk_h);
    const int nsteps = chunk_h + RI;
    const int p0 = y0 - R - E;  // source row of stream index 0 (reflected)

    const float* __restrict__ src = in + (size_t)img * (size_t)w * (size_t)h;
    const size_t img_off = (size_t)img * (size_t)w * (size_t)h;
    float* ring = s_ring[wave];

    // Lanes beyond the strip shadow its last lane (same addresses, same values): every lane runs the
    // same instruction stream and no global access sits under a branch, which keeps the compiler's
    // vmcnt bookkeeping exact and the prefetched rows really in flight.
    const int el = min(lane, sw / CPL - 1);
    const int mcol = xs + CPL * el;
    // halo: lanes < 2*RA fetch one reflected column each (the others repeat lane 0's and drop it)
    const bool has_halo = lane < 2 * RA;
    const int hl = has_halo ? lane : 0;
    const int hcol = reflect_clamp(hl < RA ? xs - RA + hl : xs + sw + (hl - RA), w);
    const int hslot = hl < RA ? hl : RA + sw + (hl - RA);
    // uniform row base (SGPR pair) + 32-bit per-lane byte offset: no 64-bit per-lane addresses to keep
    const unsigned moff = 4u * (unsigned)mcol, hoff = 4u * (unsigned)hcol;
    int dcol[CPL];   // DEC: destination column of each of this lane's source columns (-1: dropped)
#pragma unroll
    for (int e = 0; e < CPL; ++e) dcol[e] = DEC ? dec.inv_x[mcol + e] : 0;

    float tp[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) tp[k] = taps[k];

    f4v A[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) A[j] = (f4v)(0.0f);

    f4v pm[PF];
    float ph[PF];
    f4v W[NV4];
    // rows past the end of the stream are clamped to a legal row and never used
#define SIFT_STREAM_FETCH(T, U)                                                                                       \
    {                                                                                                                 \
        const char* rowp_ = reinterpret_cast<const char*>(src + (size_t)reflect_clamp(p0 + (T), h) * (size_t)w);      \
        pm[U] = *reinterpret_cast<const f4v*>(rowp_ + moff);                                                          \
        ph[U] = *reinterpret_cast<const float*>(rowp_ + hoff);                                                        \
    }
    // LDS stage of stream row T (held in pm[U]): into the ring, refill pm[U] from HBM, read the window
#define SIFT_STREAM_LDS(T, U)                                                                                         \
    {                                                                                                                 \
        float* row_ = ring + wslot * ROWF;                                                                            \
        *reinterpret_cast<f4v*>(row_ + RA + CPL * el) = pm[U];                                                          \
        if (has_halo) row_[hslot] = ph[U];                                                                            \
        __builtin_amdgcn_wave_barrier();                                                                              \
        SIFT_STREAM_FETCH((T) + PF, U)                                                                                \
        const f4v* p4_ = reinterpret_cast<const f4v*>(row_) + el;                                                     \
        _Pragma("unroll") for (int c = 0; c < NV4; ++c) W[c] = p4_[c];                                                \
        wslot = wslot + 1 == DP ? 0 : wslot + 1;                                                                      \
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) SIFT_STREAM_FETCH(u, u)

    int wslot = 0;                 // ring slot the next LDS stage writes
    int pslot = DP > 1 ? 2 : 0;    // ring slot of stream row s - R   (-R mod (R + 2))
    SIFT_STREAM_LDS(0, 0)

    // one step: row pass of stream row S from the window read a step earlier; the next row's LDS stage
    // is issued before the column pass so that its latency hides under it
#define SIFT_STREAM_STEP(S, U, STORE)                                                                                 \
    {                                                                                                                 \
        f4v m = (f4v)(0.0f);                                                                                          \
        {                                                                                                             \
            float v[NV4 * CPL];                                                                                       \
            _Pragma("unroll") for (int c = 0; c < NV4; ++c)                                                           \
                _Pragma("unroll") for (int e = 0; e < CPL; ++e) v[CPL * c + e] = W[c][e];                             \
            _Pragma("unroll") for (int k = 0; k < NT; ++k) {                                                          \
                const float tap = tp[NT - 1 - k];                                                                     \
                _Pragma("unroll") for (int e = 0; e < CPL; ++e) m[e] += tap * v[PAD + k + e];                         \
            }                                                                                                         \
        }                                                                                                             \
        f4v prev = (f4v)(0.0f);                                                                                       \
        if (DOG && (STORE)) prev = *reinterpret_cast<const f4v*>(ring + pslot * ROWF + RA + CPL * el);                  \
        __builtin_amdgcn_wave_barrier();                                                                              \
        SIFT_STREAM_LDS((S) + 1, ((U) + 1) % PF)                                                                      \
        /* The taps are symmetric (tap[j] == tap[2R-j] bit for bit: initGaussian evaluates x*x), so the product  */   \
        /* tap[j] * mid is the same float for slots j and 2R-j: one multiply serves both additions.             */   \
        {                                                                                                             \
            f4v An[NT];                                                                                               \
            _Pragma("unroll") for (int i = 0; i <= R; ++i) {                                                          \
                const f4v pr = tp[i] * m;                                                                             \
                An[i] = A[i + 1] + pr;            /* i + 1 <= R + 1 <= 2R for R >= 1 */                               \
                if (2 * R - i != i) An[2 * R - i] = (2 * R - i + 1 < NT ? A[2 * R - i + 1] : (f4v)(0.0f)) + pr;       \
            }                                                                                                         \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) A[j] = An[j];                                              \
        }                                                                                                             \
        if (STORE) {                                                                                                  \
            const int y = y0 + (S) - RI;                                                                              \
            const size_t o = img_off + (size_t)y * (size_t)w;                                                         \
            if (DOG) {                                                                                                \
                const f4v dif = A[0] - prev;                                                                          \
                __builtin_nontemporal_store((f4v)(128.0f + dif), reinterpret_cast<f4v*>(reinterpret_cast<char*>(dog + o) + moff)); \
            }                                                                                                         \
            if (DEC) {                                                                                                \
                const int jd = dec.inv_y[y];   /* wave-uniform: half of the rows are dropped */                       \
                if (jd >= 0) {                                                                                        \
                    float* drow = out + ((size_t)img * (size_t)dec.hd + (size_t)jd) * (size_t)dec.wd;                 \
                    _Pragma("unroll") for (int e = 0; e < CPL; ++e)                                                   \
                        if (dcol[e] >= 0) drow[dcol[e]] = A[0][e];   /* consecutive lanes, consecutive columns */     \
                }                                                                                                     \
            } else if (out) {   /* out == nullptr: only the DoG is wanted (wave-uniform) */                            \
                __builtin_nontemporal_store(A[0], reinterpret_cast<f4v*>(reinterpret_cast<char*>(out + o) + moff));   \
            }                                                                                                         \
        }                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        pslot = pslot + 1 == DP ? 0 : pslot + 1;                                                                      \
    }

    // run-in: RI rows that only feed the partial sums
    int s0 = 0;
#pragma unroll 1
    for (; s0 < RI; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) SIFT_STREAM_STEP(s0 + u, u, false)
    }
    // steady state: every step completes one output row (chunk_h is a multiple of PF: no tail)
#pragma unroll 1
    for (; s0 < nsteps; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) SIFT_STREAM_STEP(s0 + u, u, true)
    }
#undef SIFT_STREAM_STEP
#undef SIFT_STREAM_LDS
#undef SIFT_STREAM_FETCH
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
    if (out) out[o] = sum;
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

// Optional timing events of the launch in progress: handed to hipExtLaunchKernelGGL, which stamps them
// from the dispatch packet itself (no extra barrier packets between back-to-back launches).
static thread_local hipEvent_t t_ev_start = nullptr, t_ev_stop = nullptr;

constexpr long long kSmallTileLaunch = 4096;   // 64x64 tiles: up to 16 per CU

// Workgroups of blur_fused_kernel<R, ., TH> one launch has resident at a time: CUs x the workgroups per CU its launch bounds ask
// for.  (Round 3 asked the runtime - hipOccupancyMaxActiveBlocksPerMultiprocessor, once per thread and instantiation, in front of
// the launch: a runtime call that creates the kernel's function object beside other threads' launches.  The launch path makes
// no such call any more; the CU count is read once per device when the first context is created, sift_hip_create.)
static int g_cus[64] = {0};
int resident_cus() {
    const int cus = g_cus[(unsigned)tracked_device() % 64u];
    return cus > 0 ? cus : 256;
}
template <int R, int TH>
static int resident_workgroups() {
    constexpr int per_cu = (R <= 8 ? 3 : ((R <= 24 || (TH <= 48 && R <= 28)) ? 2 : 1));   // blur_fused_kernel's __launch_bounds__
    const int cus = g_cus[(unsigned)tracked_device() % 64u];
    return per_cu * (cus > 0 ? cus : 256);
}

template <int R, int TH>
static void launch_fused_rt(hipStream_t s, const float* in, float* out, float* dog, int w, int h, int n,
                            const float* d_taps) {
    const int tiles_x = (w + 63) / 64, tiles_y = (h + TH - 1) / TH;
    const int total = tiles_x * tiles_y * n;
    // Persistent workgroups: as many as are resident at once, so that every workgroup starts at once and fetches its next
    // tile under the arithmetic of the current one.  (More than that - 1024 in round 2 - run in a second round that begins
    // when the first ends: two tile latencies back to back and nothing prefetched.)
    const int cap = resident_workgroups<R, TH>();
    int grid = total < cap ? total : cap;
    if (grid >= 8) grid &= ~7;
    const bool aligned = (((uintptr_t)in | (uintptr_t)out | (uintptr_t)dog) & 15u) == 0;
    const int vec_ok = (w % 4 == 0 && aligned) ? 1 : 0;
    if (dog)
        hipExtLaunchKernelGGL((blur_fused_kernel<R, true, TH>), dim3((unsigned)grid), dim3(256), 0, s, t_ev_start, t_ev_stop, 0, in, out, dog, w, h,
                           tiles_x, tiles_y, total, vec_ok, d_taps);
    else
        hipExtLaunchKernelGGL((blur_fused_kernel<R, false, TH>), dim3((unsigned)grid), dim3(256), 0, s, t_ev_start, t_ev_stop, 0, in, out, dog, w, h,
                           tiles_x, tiles_y, total, vec_ok, d_taps);
}

template <int R>
static void launch_fused_r(hipStream_t s, const float* in, float* out, float* dog, int w, int h, int n,
                           const float* d_taps) {
    // Small launches (a few tiles per CU) are bound by the latency of one tile, not by throughput: 48-row tiles
    // shorten it (and let two workgroups share a CU's LDS at the largest radii) at the price of more halo rows
    // per output row, which only matters once the launch fills the chip several times over.
    const long long tiles64 = (long long)((w + 63) / 64) * ((h + 63) / 64) * n;
    if (tiles64 <= kSmallTileLaunch) return launch_fused_rt<R, 48>(s, in, out, dog, w, h, n, d_taps);
    launch_fused_rt<R, 64>(s, in, out, dog, w, h, n, d_taps);
}

constexpr int kStreamWaves = 2048;     // waves a streaming launch is cut into (8 per CU)
static int g_stream_waves = kStreamWaves;   // option "stream_waves" (process-wide; 0: tile kernel only): A/B measurements
void set_stream_waves(int v) { g_stream_waves = v < 0 ? kStreamWaves : v; }
static int stream_waves() { return g_stream_waves; }

constexpr int kMaxRadiusStream = 14;   // beyond: the 2R+1 partial sums per column no longer fit the register file

template <int R, int CPL>
static bool launch_stream_rc(hipStream_t s, const float* in, float* out, float* dog, int w, int h, int n,
                             const float* d_taps, int min_waves, const StreamDecimate* dec = nullptr) {
    const int target = stream_waves();
    if (target <= 0) return false;
    const bool aligned = (((uintptr_t)in | (dec ? 0 : (uintptr_t)out) | (uintptr_t)dog) & (4u * CPL - 1u)) == 0;
    if (!(w % CPL == 0 && aligned) || w < CPL || h < R + 1 || w < R + 1) return false;
    constexpr int SW = 64 * CPL;
    const int strips = (w + SW - 1) / SW;
    const int strip_w = (((w + strips - 1) / strips) + CPL - 1) / CPL * CPL;
    constexpr int PF = kStreamPF;
    constexpr int RI = stream_runin(R);
    int chunks = target / (n * strips);
    if (chunks < 1) chunks = 1;
    int chunk_h = (h + chunks - 1) / chunks;
    if (chunk_h < 3 * RI) chunk_h = 3 * RI;  // keep the run-in rows a minor share
    chunk_h = (chunk_h + PF - 1) / PF * PF;  // whole unrolled bodies
    if (chunk_h > h) return false;
    chunks = (h + chunk_h - 1) / chunk_h;
    const int total = n * strips * chunks;
    if (total < min_waves) return false;
    const int grid = (total + 3) / 4;
    const StreamDecimate none{nullptr, nullptr, 0, 0, nullptr};
    if (dec)
        hipExtLaunchKernelGGL((blur_stream_kernel<R, false, CPL, true>), dim3((unsigned)grid), dim3(256), 0, s, t_ev_start, t_ev_stop, 0, in,
                              out, (float*)nullptr, w, h, strips, strip_w, chunks, chunk_h, total, d_taps, *dec);
    else if (dog)
        hipExtLaunchKernelGGL((blur_stream_kernel<R, true, CPL>), dim3((unsigned)grid), dim3(256), 0, s, t_ev_start, t_ev_stop, 0, in, out, dog, w, h,
                           strips, strip_w, chunks, chunk_h, total, d_taps, none);
    else
        hipExtLaunchKernelGGL((blur_stream_kernel<R, false, CPL>), dim3((unsigned)grid), dim3(256), 0, s, t_ev_start, t_ev_stop, 0, in, out, dog, w,
                           h, strips, strip_w, chunks, chunk_h, total, d_taps, none);
    return true;
}

// 4 columns per lane while the kernel is HBM-bound (r <= 5); 2 columns per lane beyond, where the
// arithmetic per row grows and the halved register footprint buys the waves to overlap it with HBM.
template <int R>
static bool launch_stream_r(hipStream_t s, const float* in, float* out, float* dog, int w, int h, int n,
                            const float* d_taps, int min_waves, const StreamDecimate* dec = nullptr) {
    if constexpr (R > kMaxRadiusStream) {
        return false;
    } else if constexpr (R <= 5) {
        return launch_stream_rc<R, 4>(s, in, out, dog, w, h, n, d_taps, min_waves, dec);
    } else {
        return launch_stream_rc<R, 2>(s, in, out, dog, w, h, n, d_taps, min_waves, dec);
    }
}

#define SIFT_FUSED_CASE(R) \
    case R:                \
        if (launch_stream_r<R>(s, in, out, dog, w, h, n, d_taps, min_waves)) return; \
        launch_fused_r<R>(s, in, out, dog, w, h, n, d_taps); \
        return;

void launch_blur(hipStream_t s, bool fused, const float* in, float* tmp, float* out, float* dog, int w,
                 int h, int n, const float* d_taps, int radius, int min_waves, hipEvent_t ev_start, hipEvent_t ev_stop) {
    struct Scope {
        Scope(hipEvent_t a, hipEvent_t b) { t_ev_start = a; t_ev_stop = b; }
        ~Scope() { t_ev_start = t_ev_stop = nullptr; }
    } scope(ev_start, ev_stop);
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
    // two-pass fallback: two kernels, bracketed by ordinary event records
    if (ev_start) (void)hipEventRecord(ev_start, s);
    const dim3 grid((unsigned)((w + 255) / 256), (unsigned)h, (unsigned)n);
    const size_t shm = sizeof(float) * (size_t)(2 * radius + 1);
    hipLaunchKernelGGL(gauss_row_generic, grid, dim3(256), shm, s, in, tmp, w, h, d_taps, radius);
    if (dog)
        hipLaunchKernelGGL(gauss_col_generic<true>, grid, dim3(256), shm, s, (const float*)tmp, in, out, dog, w,
                           h, d_taps, radius);
    else
        hipLaunchKernelGGL(gauss_col_generic<false>, grid, dim3(256), shm, s, (const float*)tmp, in, out, dog,
                           w, h, d_taps, radius);
    if (ev_stop) (void)hipEventRecord(ev_stop, s);
}

#define SIFT_REDUCE_CASE(R) \
    case R:                 \
        return launch_stream_r<R>(s, in, dst, nullptr, w, h, n, d_taps, min_waves, &dec);

// Blur + nearest-neighbour decimation in one pass (streaming kernel only).  false: the caller runs the blur into a
// temporary and the resampling kernel after it.
bool launch_blur_reduce(hipStream_t s, const float* in, float* dst, int w, int h, int wd, int hd, int n, const float* d_taps,
                        int radius, const int* d_inv_x, const int* d_inv_y, float* d_dump, int min_waves, hipEvent_t ev_start,
                        hipEvent_t ev_stop) {
    struct Scope {
        Scope(hipEvent_t a, hipEvent_t b) { t_ev_start = a; t_ev_stop = b; }
        ~Scope() { t_ev_start = t_ev_stop = nullptr; }
    } scope(ev_start, ev_stop);
    const StreamDecimate dec{d_inv_x, d_inv_y, wd, hd, d_dump};
    switch (radius) {
        SIFT_REDUCE_CASE(1) SIFT_REDUCE_CASE(2) SIFT_REDUCE_CASE(3) SIFT_REDUCE_CASE(4) SIFT_REDUCE_CASE(5)
        SIFT_REDUCE_CASE(6) SIFT_REDUCE_CASE(7) SIFT_REDUCE_CASE(8) SIFT_REDUCE_CASE(9) SIFT_REDUCE_CASE(10)
        SIFT_REDUCE_CASE(11) SIFT_REDUCE_CASE(12) SIFT_REDUCE_CASE(13) SIFT_REDUCE_CASE(14)
    }
    return false;
}

void launch_resample(hipStream_t s, const float* src, float* dst, int ws, int hs, int wd, int hd, int n,
                     const int* d_lutx, const int* d_luty) {
    const dim3 grid((unsigned)((wd + 255) / 256), (unsigned)hd, (unsigned)n);
    hipLaunchKernelGGL(resample_lut_kernel, grid, dim3(256), 0, s, src, dst, ws, hs, wd, hd, d_lutx, d_luty);
}

// 8-bit samples -> the integer-valued floats vigra::importImage hands to Sift::calculate (/root/reference/main.cpp:52-54):
// a thread widens 16 consecutive bytes (one 16-byte load, four 16-byte stores)
__global__ void widen_u8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, size_t count) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i + 16 <= count && (((uintptr_t)in | (uintptr_t)out) & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(in + i);
        const unsigned wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *reinterpret_cast<float4*>(out + i + 4 * k) = make_float4((float)(wd[k] & 255u), (float)((wd[k] >> 8) & 255u),
                                                                      (float)((wd[k] >> 16) & 255u), (float)(wd[k] >> 24));
    } else {
        for (size_t k = i; k < count && k < i + 16; ++k) out[k] = (float)in[k];
    }
}

void launch_widen_u8(hipStream_t s, const uint8_t* in, float* out, size_t count) {
    const size_t threads = (count + 15) / 16;
    hipLaunchKernelGGL(widen_u8_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, in, out, count);
}

void launch_dog(hipStream_t s, const float* lower, const float* higher, float* out, size_t count) {
    const unsigned grid = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(dog_kernel, dim3(grid), dim3(256), 0, s, lower, higher, out, count);
}

// The runtime builds a translation unit's device code on the first launch of any of its kernels, and two host threads that make
// their first launches at the same time (several contexts, one thread each) were seen to crash inside that step
// (tools/asan_example.sh: SEGV below hipLaunchKernel).  sift_hip_create touches every unit once, under a lock.
__global__ void tu_probe_pyramid_kernel() {}
void tu_touch_pyramid(hipStream_t s) {
    int cus = 0;
    const int dev = tracked_device();
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) g_cus[(unsigned)dev % 64u] = cus;
    (void)hipGetLastError();
    hipLaunchKernelGGL(tu_probe_pyramid_kernel, dim3(1), dim3(1), 0, s);
}

}  // namespace sift_hip
