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
#include "blur_stream.h"   // reflect_clamp, blur_stream_kernel

#pragma clang fp contract(off)

namespace sift_hip {

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

// Rows in pairs (blur_stream.h: blur_stream2_kernel; round 6): the level launches of radius 6 .. 14 that write a Gaussian level
// only - no DoG, no decimation -, i.e. every streaming launch of the default plan beyond the first two levels.
// waves such a launch is cut into: one round of the slots its registers leave - three waves per SIMD up to radius 10 (3072 slots;
// 2880 waves for 32 x 1080p), two beyond (tools/probe/blur_probe.hip, profiles/r06_blur_probe.txt: R 10 alone 172 us as columns
// packed, 149 us at 1920 waves, 147 at 2880, 157 at 2400 - a round and a sixth)
constexpr int stream2_waves(int r) { return stream2_occ(r) >= 3 ? 3072 : 2048; }
template <int R>
static bool launch_stream2_r(hipStream_t s, const float* in, float* out, int w, int h, int n, const float* d_taps, int min_waves) {
    const int target = stream_waves() == kStreamWaves ? stream2_waves(R) : stream_waves();
    if (target <= 0) return false;
    const bool aligned = (((uintptr_t)in | (uintptr_t)out) & 7u) == 0;
    if (!(w % 2 == 0 && aligned) || w < 2 || h < R + 1 || w < R + 1) return false;
    if ((long long)w * h * 4 > 0x7fffffffLL) return false;   // an image's rows are addressed by 32-bit byte offsets (buffer descriptors)
    constexpr int SW = 128;
    const int strips = (w + SW - 1) / SW;
    const int strip_w = (((w + strips - 1) / strips) + 1) / 2 * 2;
    constexpr int RI = stream_runin(R);
    int chunks = target / (n * strips);
    if (chunks < 1) chunks = 1;
    int chunk_h = (h + chunks - 1) / chunks;
    if (chunk_h < 3 * RI) chunk_h = 3 * RI;  // keep the run-in rows a minor share
    chunk_h = (chunk_h + 3) / 4 * 4;         // whole unrolled bodies (two row pairs)
    if (chunk_h > h) return false;
    chunks = (h + chunk_h - 1) / chunk_h;
    const int total = n * strips * chunks;
    if (total < min_waves) return false;
    hipExtLaunchKernelGGL((blur_stream2_kernel<R, 1>), dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, t_ev_start, t_ev_stop, 0, in, out, w, h, strips,
                          strip_w, chunks, chunk_h, total, d_taps);
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
        if (!dog && out && !dec && launch_stream2_r<R>(s, in, out, w, h, n, d_taps, min_waves)) return true;
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
