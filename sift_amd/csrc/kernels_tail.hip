// The pyramid's tail as ONE launch (round 5): every image's small octaves - level blurs with their DoGs and the reductions
// between them (Sift::_createDOGs' level loop, /root/reference/sift.cpp:381-417; alg::convolveWithGauss / dog /
// reduceToNextLevel, algorithms.cpp:10-36,52-64) - walked end to end by ONE 1024-thread workgroup per image.
//
// Why: for the whole batch these octaves are a few tiles per CU and launch.  As launches of their own they are bound by the
// latency of a tile and of a launch boundary (eight launches, 0.04 - 0.18 of the HBM peak, ~250 us of a 1.09 ms pyramid in
// round 4) while the chip stands idle; as a tile chain with inter-workgroup flags (round 3's blur_chain_kernel) they were
// slower still.  Here a dependency never leaves its workgroup: a level is complete when the workgroup's threads have passed
// a barrier, the next level reads it back through the CU's own L1 / the XCD's L2 (workgroup scope: no cache maintenance, no
// flags), and the kernel is bound by instruction issue on the ~n CUs it occupies - 2 (2R+1) multiply-adds per pixel and level, the
// reference's own count - while the rest of the chip runs the partner batch's descriptors and this batch's extremum scans of
// the large octaves (context.cpp: option "tail_kernel", the stream it is launched on, what waits for it).
//
// One op (TailOp) = one convolveWithGauss of a W x H level, in bands of BR output rows:
//   * source rows enter a staging area in LDS with their reflected halo columns (RA = R rounded up to 4 each side);
//   * row pass: a thread forms 4 consecutive outputs of a row from a window of 16-byte LDS reads - ascending order from
//     0.0f, one rounding per multiply and per add, as kernels_pyramid.hip - into a RING of row-pass rows in LDS that
//     slides down the image (BR + 2R rows: every source row is row-passed exactly once); the rows the column pass reads
//     beyond the image's top and bottom are copies of their reflections (reflect(p) = -p, 2(h-1)-p), made in LDS;
//   * column pass: a thread forms 4 columns x 2 rows from 2R+2 ring rows (conflict-free 16-byte reads), subtracts the
//     source level for the DoG (128.0f + (new - prev)) and stores 16 bytes per level;
//   * a reduction (kind 3) row-passes every source row and runs the column pass only at the pixels the nearest-neighbour
//     decimation keeps (index maps built on the host with Vigra's accumulated-double rule).
// Bit-exactness contract: kernels_pyramid.hip's.  -ffp-contract=off.
#include "common.h"
#include "lds_tile.h"
#include "tail_plan.h"

#pragma clang fp contract(off)

namespace sift_hip {

namespace {

#ifndef SIFT_TAIL_THREADS
#define SIFT_TAIL_THREADS 1024
#endif
constexpr int kTailThreads = SIFT_TAIL_THREADS;
constexpr int kTailWaves = kTailThreads / 64;

// Level memory is addressed as GLOBAL (address space 1), not through generic pointers: the level bases come out of the argument
// block, where the compiler cannot see their address space, and a FLAT load counts on the LDS counter too - every
// `s_waitcnt lgkmcnt(0)` of the LDS reads below would then wait for the source rows that are meant to stay in flight.
typedef __attribute__((address_space(1))) const float gcf;
typedef __attribute__((address_space(1))) float gf;
typedef float f4n __attribute__((ext_vector_type(4)));   // (float4 is a class: no objects of it in a named address space)
typedef __attribute__((address_space(1))) const f4n gcf4;
typedef __attribute__((address_space(1))) f4n gf4;
typedef __attribute__((address_space(1))) const int gci;

__device__ __forceinline__ int tail_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T* tail_uni_ptr(T* p) {
    const uintptr_t a = (uintptr_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return (T*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
}

__device__ __forceinline__ int tail_reflect(int p, int n) {
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * (n - 1) - p : p;
    return p < 0 ? 0 : (p >= n ? n - 1 : p);   // columns of padding lanes may still be out of range: keep them legal
}

__device__ __forceinline__ void tail_lds_read4(const unsigned (&ad)[4], float4 (&m)[4]) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3])
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3])
                 : "memory");
}
__device__ __forceinline__ void tail_lds_read2(const unsigned (&ad)[4], float4 (&m)[4]) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(m[0]), "=&v"(m[1])
                 : "v"(ad[0]), "v"(ad[1])
                 : "memory");
}

// Staging of source rows [r0, r0 + nr) (all inside the image): logical columns -RA .. WP + RA - 1, reflected, as 16-byte units
// e = row * SW4 + u.  The first kTailPre units of a thread are FETCHED early - into registers, before the column pass of the
// band in front - and dropped into LDS when the staging area is free again; a level too wide for that loads the rest then.
constexpr int kTailPre = 3;
struct TailStage {
    float4 v[kTailPre];
};
__device__ __forceinline__ float4 tail_stage_unit(gcf* __restrict__ src, int w, int r0, int e, int RA, int SW4, bool vec_ok) {
    const int row = e / SW4, u = e - row * SW4;
    const int x0 = 4 * u - RA;
    gcf* __restrict__ rp = src + (size_t)(r0 + row) * (size_t)w;
    float4 v;
    if (vec_ok && x0 >= 0 && x0 + 3 < w) {
        const f4n t = *reinterpret_cast<gcf4*>(rp + x0);
        v = make_float4(t.x, t.y, t.z, t.w);
    } else {
        v.x = rp[tail_reflect(x0 + 0, w)];
        v.y = rp[tail_reflect(x0 + 1, w)];
        v.z = rp[tail_reflect(x0 + 2, w)];
        v.w = rp[tail_reflect(x0 + 3, w)];
    }
    return v;
}
__device__ __forceinline__ void tail_stage_fetch(TailStage& st, gcf* __restrict__ src, int w, int r0, int nr, int RA, int SW4, bool vec_ok) {
    const int total = nr * SW4;
#pragma unroll
    for (int i = 0; i < kTailPre; ++i) {
        const int e = (int)threadIdx.x + kTailThreads * i;
        if (e < total) st.v[i] = tail_stage_unit(src, w, r0, e, RA, SW4, vec_ok);
    }
}
__device__ __forceinline__ void tail_stage_commit(const TailStage& st, gcf* __restrict__ src, int w, int r0, int nr, int RA, int SW4, bool vec_ok,
                                                  float4* __restrict__ stage4) {
    const int total = nr * SW4;
#pragma unroll
    for (int i = 0; i < kTailPre; ++i) {
        const int e = (int)threadIdx.x + kTailThreads * i;
        if (e < total) stage4[e] = st.v[i];
    }
#pragma unroll 1
    for (int e = (int)threadIdx.x + kTailThreads * kTailPre; e < total; e += kTailThreads) stage4[e] = tail_stage_unit(src, w, r0, e, RA, SW4, vec_ok);
}

// Work is dealt to WAVES: a unit is one row (row pass) or one pair of rows (column pass) x a run of 64 column groups, so a
// unit's rows - staging row, ring slots, their wrap-around - are wave-uniform and live in scalar registers; a lane only adds
// its column offset.  (Dealt to threads, every lane carried its own row: a division and three integer operations per LDS read.)

// row pass of staged rows 0 .. nr-1 into ring slots of logical rows r0 .. (slot of logical row r = (r + R) % MR)
template <int R>
__device__ __forceinline__ void tail_row_pass(const float4* __restrict__ stage4, int SW4, int G, int C, int nr, int r0, int MR,
                                              float4* __restrict__ ring4, const float (&tp)[R + 1]) {
    constexpr int RA = (R + 3) & ~3;
    constexpr int PAD = RA - R;
    constexpr int NT = 2 * R + 1;
    constexpr int NV = PAD + 4 + 2 * R;
    constexpr int NV4 = (NV + 3) / 4;
    const int lane = (int)threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int units = nr * C;
#pragma unroll 1
    for (int u = wave; u < units; u += kTailWaves) {
        const int row = u / C, c = u - row * C;
        const int g = 64 * c + lane;
        const int gc = min(g, G - 1);   // lanes past the row's end shadow its last group and store nothing
        float4 f4[NV4];
        lds_read_window<NV4>(stage4 + row * SW4 + gc, f4);
        float v[NV4 * 4];
#pragma unroll
        for (int q = 0; q < NV4; ++q) {
            v[4 * q + 0] = f4[q].x;
            v[4 * q + 1] = f4[q].y;
            v[4 * q + 2] = f4[q].z;
            v[4 * q + 3] = f4[q].w;
        }
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const float tap = tp[k <= R ? k : 2 * R - k];   // tap[2R - k] == tap[k], bit for bit
            a0 += tap * v[PAD + k];
            a1 += tap * v[PAD + k + 1];
            a2 += tap * v[PAD + k + 2];
            a3 += tap * v[PAD + k + 3];
        }
        const int slot = (r0 + row + R) % MR;
        if (g < G) ring4[slot * G + g] = make_float4(a0, a1, a2, a3);
    }
}

// logical rows [a, b) outside the image: copies of their reflections, which the ring still holds
__device__ __forceinline__ void tail_virtual_rows(float4* __restrict__ ring4, int G, int MR, int R, int h, int a, int b) {
    const int total = (b - a) * G;
#pragma unroll 1
    for (int t = (int)threadIdx.x; t < total; t += kTailThreads) {
        const int i = t / G, g = t - i * G;
        const int r = a + i;
        const int q = r < 0 ? -r : 2 * (h - 1) - r;
        ring4[((r + R) % MR) * G + g] = ring4[((q + R) % MR) * G + g];
    }
}

// column pass of output rows [y0, y1): a wave takes two rows x 64 column groups; level and DoG stores
template <int R>
__device__ __forceinline__ void tail_col_pass(const float4* __restrict__ ring4, int G, int C, int MR, int y0, int y1, int w, bool vec_ok,
                                              gcf* __restrict__ src, gf* __restrict__ dst, gf* __restrict__ dog, const float (&tp)[R + 1]) {
    constexpr int PY = 2;
    constexpr int NK = PY + 2 * R;
    const int lane = (int)threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int units = ((y1 - y0 + PY - 1) / PY) * C;
    const unsigned row_bytes = 16u * (unsigned)G, ring_bytes = row_bytes * (unsigned)MR;
#pragma unroll 1
    for (int u = wave; u < units; u += kTailWaves) {
        const int rg = u / C, c = u - rg * C;
        const int g = 64 * c + lane;
        const int gc = min(g, G - 1);
        const int y = y0 + rg * PY;
        float4 acc[PY];
#pragma unroll
        for (int i = 0; i < PY; ++i) acc[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const unsigned lane_addr = lds_addr(ring4) + 16u * (unsigned)gc;
        unsigned s_off = (unsigned)(y % MR) * row_bytes;   // logical row y - R; wave-uniform, wraps at the ring's end
        // 2R + 2 ring rows, four at a time: the reads and their wait are one asm statement (the optimiser would otherwise
        // hoist every read of the unrolled loop to the top: 4 registers per tap)
#pragma unroll
        for (int k0 = 0; k0 < NK; k0 += 4) {
            float4 m[4];
            unsigned ad[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ad[q] = lane_addr + s_off;
                if (k0 + q < NK) {
                    s_off += row_bytes;
                    s_off = s_off == ring_bytes ? 0u : s_off;
                }
            }
            if (k0 + 4 <= NK) tail_lds_read4(ad, m);
            else tail_lds_read2(ad, m);   // (NK is even)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + q;
                if (k >= NK) break;
#pragma unroll
                for (int i = 0; i < PY; ++i) {
                    if (k - i >= 0 && k - i <= 2 * R) {
                        const float tap = tp[k - i <= R ? k - i : 2 * R - (k - i)];
                        acc[i].x += tap * m[q].x;
                        acc[i].y += tap * m[q].y;
                        acc[i].z += tap * m[q].z;
                        acc[i].w += tap * m[q].w;
                    }
                }
            }
        }
        // (a use the optimiser cannot move: otherwise it sinks each row's sums into the branch that stores them and keeps
        // every ring row it has read alive - in scratch - until then)
#pragma unroll
        for (int i = 0; i < PY; ++i) asm volatile("" : "+v"(acc[i].x), "+v"(acc[i].y), "+v"(acc[i].z), "+v"(acc[i].w));
        const int x = 4 * g;
#pragma unroll
        for (int i = 0; i < PY; ++i) {
            const int yy = y + i;
            if (yy >= y1 || g >= G || x >= w) continue;
            const size_t o = (size_t)yy * (size_t)w + (size_t)x;
            if (vec_ok && x + 3 < w) {
                if (dog) {
                    const f4n prev = *reinterpret_cast<gcf4*>(src + o);
                    const float dx = acc[i].x - prev.x, dy = acc[i].y - prev.y, dz = acc[i].z - prev.z, dw = acc[i].w - prev.w;
                    *reinterpret_cast<gf4*>(dog + o) = (f4n){128.0f + dx, 128.0f + dy, 128.0f + dz, 128.0f + dw};
                }
                if (dst) *reinterpret_cast<gf4*>(dst + o) = (f4n){acc[i].x, acc[i].y, acc[i].z, acc[i].w};
            } else {
                const float av[4] = {acc[i].x, acc[i].y, acc[i].z, acc[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (x + j < w) {
                        if (dog) {
                            const float dif = av[j] - src[o + j];
                            dog[o + j] = 128.0f + dif;
                        }
                        if (dst) dst[o + j] = av[j];
                    }
            }
        }
    }
}

// reduction: column pass at the kept pixels only - destination rows [j0, j1) (their source rows lie in the current band),
// destination columns 4 per thread, gathered from the ring through the column map
__device__ __forceinline__ void tail_col_pass_kept(const float* __restrict__ ring, int WP, int MR, int R, int j0, int j1, int wd,
                                                   gci* __restrict__ lutx, gci* __restrict__ luty, gf* __restrict__ dst,
                                                   const float* __restrict__ taps) {
    const int NT = 2 * R + 1;
    const int GD = (wd + 3) / 4;
    const int total = (j1 - j0) * GD;
#pragma unroll 1
    for (int t = (int)threadIdx.x; t < total; t += kTailThreads) {
        const int jr = t / GD, g = t - jr * GD;
        const int j = j0 + jr;
        const int y = luty[j];
        int xs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) xs[e] = lutx[min(4 * g + e, wd - 1)];
        float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        int slot = y % MR;   // logical row y - R
#pragma unroll 4
        for (int k = 0; k < NT; ++k) {
            const float* __restrict__ row = ring + slot * WP;
            slot = slot + 1 == MR ? 0 : slot + 1;
            const float tap = taps[NT - 1 - k];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += tap * row[xs[e]];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * g + e < wd) dst[(size_t)j * (size_t)wd + (size_t)(4 * g + e)] = a[e];
    }
}

// LDS of the kernel: the ring of row-pass rows, then the staging rows.  At namespace scope because every radius is a function of
// its own (not inlined: one register allocation per radius instead of one for all 32), and each must see the array as LDS.
__shared__ __attribute__((aligned(16))) float g_tail_lds[kTailLdsFloats];

template <int R>
__device__ __noinline__ void tail_op(const TailOp& op, int img, const float* __restrict__ taps_all, const int* __restrict__ luts) {
    float* smem = g_tail_lds;
    constexpr int RA = (R + 3) & ~3;
    // Arguments of a function that is not a kernel arrive in vector registers and count as divergent; everything here is the
    // same for the whole workgroup, and saying so (readfirstlane) puts sizes, row offsets, ring slots and taps into scalar
    // registers and their arithmetic on the scalar unit.
    img = tail_uni(img);
    const int w = tail_uni(op.w), h = tail_uni(op.h), wd = tail_uni(op.wd), hd = tail_uni(op.hd);
    const int WP = (w + 3) & ~3, G = WP / 4, C = (G + 63) / 64;
    const int SW4 = (WP + 2 * RA) / 4;
    const int BR = tail_uni(op.band);
    const int MR = BR + 2 * R;
    const float* __restrict__ taps = tail_uni_ptr(taps_all) + tail_uni(op.tap_off);
    gcf* __restrict__ src = (gcf*)tail_uni_ptr(op.src) + (size_t)img * (size_t)w * (size_t)h;
    const bool reduce = tail_uni(op.kind) == 3;
    float* const dst_g = tail_uni_ptr(op.dst);
    float* const dog_g = tail_uni_ptr(op.dog);
    gf* __restrict__ dst = dst_g ? (gf*)dst_g + (size_t)img * (size_t)wd * (size_t)hd : nullptr;
    gf* __restrict__ dog = dog_g ? (gf*)dog_g + (size_t)img * (size_t)w * (size_t)h : nullptr;
    const bool vec_ok = (w % 4 == 0) && ((((uintptr_t)src | (uintptr_t)dst | (uintptr_t)dog) & 15u) == 0);
    float4* ring4 = reinterpret_cast<float4*>(smem);
    float4* stage4 = ring4 + (size_t)MR * (size_t)G;
    gci* __restrict__ lutx = (gci*)tail_uni_ptr(luts) + tail_uni(op.lut_x);
    gci* __restrict__ luty = (gci*)tail_uni_ptr(luts) + tail_uni(op.lut_y);

    // The taps are symmetric bit for bit (initGaussian evaluates x * x), so R + 1 wave-uniform values - scalar registers -
    // serve both passes: tap[k] for k <= R, tap[2R - k] beyond.
    float tp[R + 1];
#pragma unroll
    for (int k = 0; k <= R; ++k) tp[k] = __builtin_bit_cast(float, tail_uni(__builtin_bit_cast(int, taps[k])));

    int next = 0;        // next source row to row-pass
    int have = 0;        // logical rows [.., have) are in the ring (rows >= h as copies)
    int jn = 0;          // reduction: next destination row
    // The batch of source rows that is row-passed next: the rows band `yb` (or the first band behind it that needs any) still
    // lacks, at most BR of them - never past what that band reads, the ring holds no more (band 0 asks for BR + R rows: a batch
    // of BR, then one of R; every later band for BR).
    auto plan_batch = [&](int nxt, int yb, int& nr) {
        for (; yb < h; yb += BR) {
            const int rh = min(yb + BR + R, h);
            if (nxt < rh) { nr = min(BR, rh - nxt); return true; }
        }
        nr = 0;
        return false;
    };
    TailStage st;
    int pre_nr = 0;
    plan_batch(0, 0, pre_nr);
    tail_stage_fetch(st, src, w, 0, pre_nr, RA, SW4, vec_ok);   // the first batch sets out
    for (int y0 = 0; y0 < h; y0 += BR) {
        const int y1 = min(y0 + BR, h);
        const int need = min(y0 + BR + R, h + R);   // logical rows below `need` must be in the ring
        const int real_hi = min(need, h);
        const bool row_passed = next < real_hi;
        while (next < real_hi) {
            const int nb = pre_nr;   // == min(BR, real_hi - next): the batch fetched ahead is the batch used
            tail_stage_commit(st, src, w, next, nb, RA, SW4, vec_ok, stage4);
            __syncthreads();   // staging complete; the previous band's column pass is over (the ring may be overwritten)
            tail_row_pass<R>(stage4, SW4, G, C, nb, next, MR, ring4, tp);
            next += nb;
            // the following batch's rows set out now and land while this band's column pass runs
            if (plan_batch(next, y0, pre_nr)) tail_stage_fetch(st, src, w, next, pre_nr, RA, SW4, vec_ok);
            __syncthreads();
        }
        // (a band that needed no new source row has passed no barrier yet: the copies below overwrite ring rows the previous
        // band's column pass may still be reading)
        if (!row_passed && y0 > 0 && need > max(have, next)) __syncthreads();
        have = max(have, min(next, h));
        bool copied = false;
        if (y0 == 0) {   // rows -R .. -1 <- rows R .. 1
            tail_virtual_rows(ring4, G, MR, R, h, -R, 0);
            copied = true;
        }
        if (need > have) {   // rows h .. <- rows h-2 ..
            tail_virtual_rows(ring4, G, MR, R, h, have, need);
            have = need;
            copied = true;
        }
        if (copied) __syncthreads();
        if (!reduce) {
            tail_col_pass<R>(ring4, G, C, MR, y0, y1, w, vec_ok, src, dst, dog, tp);
        } else {
            int j1 = jn;
            while (j1 < hd && luty[j1] < y1) ++j1;   // the map is strictly increasing
            tail_col_pass_kept(smem, WP, MR, R, jn, j1, wd, lutx, luty, dst, taps);
            jn = j1;
        }
    }
    // the next op reads this level through the CU's L1 / the XCD's L2: workgroup scope, a barrier (and its fence) suffices
    __syncthreads();
}

#define SIFT_TAIL_CASE(R) \
    case R:               \
        tail_op<R>(op, img, taps, luts); \
        break;

}  // namespace

__global__ __launch_bounds__(kTailThreads) void pyramid_tail_kernel(TailPlan plan, const float* __restrict__ taps, const int* __restrict__ luts) {
    const int img = (int)blockIdx.x;
    for (int i = 0; i < plan.n_ops; ++i) {
        const TailOp& op = plan.op[i];
#ifdef SIFT_TAIL_ONLY   // (register reports of one radius: hipcc -DSIFT_TAIL_ONLY=19 -Rpass-analysis=kernel-resource-usage)
        tail_op<SIFT_TAIL_ONLY>(op, img, taps, luts);
        continue;
#endif
        switch (op.radius) {
            SIFT_TAIL_CASE(1) SIFT_TAIL_CASE(2) SIFT_TAIL_CASE(3) SIFT_TAIL_CASE(4) SIFT_TAIL_CASE(5) SIFT_TAIL_CASE(6)
            SIFT_TAIL_CASE(7) SIFT_TAIL_CASE(8) SIFT_TAIL_CASE(9) SIFT_TAIL_CASE(10) SIFT_TAIL_CASE(11) SIFT_TAIL_CASE(12)
            SIFT_TAIL_CASE(13) SIFT_TAIL_CASE(14) SIFT_TAIL_CASE(15) SIFT_TAIL_CASE(16) SIFT_TAIL_CASE(17) SIFT_TAIL_CASE(18)
            SIFT_TAIL_CASE(19) SIFT_TAIL_CASE(20) SIFT_TAIL_CASE(21) SIFT_TAIL_CASE(22) SIFT_TAIL_CASE(23) SIFT_TAIL_CASE(24)
            SIFT_TAIL_CASE(25) SIFT_TAIL_CASE(26) SIFT_TAIL_CASE(27) SIFT_TAIL_CASE(28) SIFT_TAIL_CASE(29) SIFT_TAIL_CASE(30)
            SIFT_TAIL_CASE(31) SIFT_TAIL_CASE(32)
        }
    }
}

// Rows per band of a W-wide level at radius r: as many as give every thread one column-pass task (4 columns x 2 rows) and
// fit the ring (band + 2r rows) and the staging area (band rows with their halo columns) into the kernel's LDS; 0: no fit.
int tail_band_rows(int w, int h, int radius) {
    if (radius < 1 || radius > kMaxRadiusFused || w < radius + 1 || h < radius + 1) return 0;
    const int WP = (w + 3) & ~3, G = WP / 4, RA = (radius + 3) & ~3;
    int band = 2 * (kTailThreads / G);
    if (band < 2) band = 2;
    if (band > 64) band = 64;
    if (band > ((h + 1) & ~1)) band = (h + 1) & ~1;
    for (; band >= 2; band -= 2) {
        const long long floats = (long long)(band + 2 * radius) * WP + (long long)band * (WP + 2 * RA);
        if (floats <= kTailLdsFloats) return band;
    }
    return 0;
}

void launch_pyramid_tail(hipStream_t s, const TailPlan& plan, int n_images, const float* d_taps, const int* d_luts, hipEvent_t ev_start,
                         hipEvent_t ev_stop) {
    hipExtLaunchKernelGGL(pyramid_tail_kernel, dim3((unsigned)n_images), dim3(kTailThreads), 0, s, ev_start, ev_stop, 0, plan, d_taps, d_luts);
}

// (the runtime builds a translation unit's device code on its first launch: sift_hip_create touches every unit once, under a lock)
__global__ void tu_probe_tail_kernel() {}
void tu_touch_tail(hipStream_t s) { hipLaunchKernelGGL(tu_probe_tail_kernel, dim3(1), dim3(1), 0, s); }

}  // namespace sift_hip
