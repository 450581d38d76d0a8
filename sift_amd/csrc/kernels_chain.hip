// Level chain of the small octaves in ONE launch (round 3; option "chain_from" = first octave, OFF by default - see the end of
// this comment for the measurement).
//
// From octave 2 on a level of the bench batch is a few tiles per CU: its launch (blur_fused_kernel, kernels_pyramid.hip) is
// bound by the latency of one tile and by the launch boundary before the next level, which needs it (Sift::_createDOGs,
// /root/reference/sift.cpp:388-411: every level is alg::convolveWithGauss of the level below, algorithms.cpp:10-22; the next
// octave starts from alg::reduceToNextLevel of the level below the top, algorithms.cpp:24-36).  Those dependencies hold PER
// IMAGE, and a launch boundary orders all 32 images at once.  This kernel takes the whole rest of the pyramid - every level
// blur (+ DoG, algorithms.cpp:52-64) and every reduction from a given octave on - as one list of work items
// (stage, image, 64 x 48 tile):
//   * the items are dealt to eight queues (image mod 8: a queue is served first by the workgroups of one XCD, ids equal mod 8,
//     so an image's levels stay near that XCD), each queue in an order in which every item comes after the items it needs;
//   * a workgroup draws the next item of its queue with one atomic add, waits until the stage the item reads from is complete
//     FOR THAT IMAGE (a counter per stage and image), runs the tile - source tile + reflected halo through registers into LDS,
//     row pass LDS -> LDS, column pass LDS -> registers, DoG from the tile still in LDS, exactly the arithmetic of
//     blur_fused_kernel: ascending-order sums from 0.0f, one rounding per multiply and per add - stores it and adds one to its
//     own stage's counter;
//   * an item is only ever drawn by a RUNNING workgroup and everything it waits for was drawn before it, so the launch makes
//     progress whatever share of the chip it is given (no assumption that all workgroups are resident); a queue that runs dry
//     lets its workgroups help the others;
//   * the top level of an octave feeds nothing but its own DoG: its items are spread between the next octave's stages, where
//     they fill the slots the (smaller) next octave leaves empty.
// A reduction is the same tile body storing only the pixels the nearest-neighbour decimation keeps (inverse index maps).
//
// Visibility of a level between workgroups (they may sit on different XCDs, whose L2s do not snoop each other for ordinary
// accesses): chain_mode 1 makes every load and store of a level a relaxed atomic of AGENT scope (8-byte halves), which the
// hardware keeps coherent across the device by itself; the writer waits for its stores (s_waitcnt) before it counts the tile.
// chain_mode 0 uses ordinary accesses between an agent-scope acquire fence and a release fence per tile - the textbook form,
// and 3.4x slower (923 against 270 us: each fence writes back / invalidates the whole L2).  chain_mode 2 is mode 1 with every draw going to the next
// queue, i.e. the tiles of one image spread over all XCDs (tests).  All three are bit-identical to the per-level launches on
// alternating batches (tests/test_gpu_parity.py::test_level_chain_on_alternating_batches).
//
// Measured on the bench batch (32 x 1080p, octaves 2 - 3: 7 stages, 7296 tiles; tools/chain_trace.sh): 268 us against 218 us
// for the eight launches it replaces (chain_from = 3: 75 against 66 us).  A tile takes 12 - 13 us (loads 3.2, row pass 4.5,
// column pass 1.6, stores 2, counters 1) x 14 tiles per workgroup; the waits are 6 us per workgroup in total - the launches
// were NOT losing their time at the boundaries, and they keep the next tile's loads in flight under the arithmetic
// (blur_fused_kernel) and evaluate a reduction at the kept pixels only (blur_reduce_kernel).  A step of the bench takes 3.30
// against 2.85 ms with it.  Two things that cost far more before they were found: the counters of all queues and images in ONE
// 128-byte line (512 workgroups' atomics serialised on it: a draw took 6 - 10 us, the launch 710 us - every counter now has a
// line of its own), and the taps read through a pointer the compiler had to assume aliased by the kernel's stores (vector
// loads and a wait inside both passes - they are read through the constant address space now: SGPRs).
#include <hip/hip_ext.h>

#include "common.h"
#include "lds_tile.h"

#pragma clang fp contract(off)

namespace sift_hip {

namespace {

__device__ __forceinline__ int chain_reflect(int p, int n) {
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * (n - 1) - p : p;
    return p < 0 ? 0 : (p >= n ? n - 1 : p);   // lanes that only feed outputs outside the image stay legal
}

constexpr int kChainTH = 48, kChainTW = 64;
constexpr int kChainMaxR = 27;
constexpr int chain_ra(int r) { return (r + 3) & ~3; }
constexpr int kChainSrcFloats = (kChainTH + 2 * kChainMaxR) * (kChainTW + 2 * chain_ra(kChainMaxR));
constexpr int kChainMidFloats = (kChainTH + 2 * kChainMaxR) * kChainTW;

// Loads and stores of the levels.  SCOPED: every access is a relaxed atomic of agent scope (8-byte halves of a lane's 16 bytes):
// it bypasses the CU's vector cache and is coherent across the device by itself, so a level written by one workgroup can be read
// by another after nothing but the writer's s_waitcnt and the completion counter - no buffer_wbl2 / buffer_inv of the whole L2
// per tile, which is what agent-scope fences around ordinary accesses cost (measured: the chain 4x slower than the launches).
template <bool SCOPED>
__device__ __forceinline__ float4 chain_ld4(const float* p) {
    if constexpr (SCOPED) {
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
        const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b),
                           __uint_as_float((unsigned)(b >> 32)));
    } else {
        return *reinterpret_cast<const float4*>(p);
    }
}
template <bool SCOPED>
__device__ __forceinline__ float chain_ld1(const float* p) {
    if constexpr (SCOPED)
        return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else
        return *p;
}
template <bool SCOPED>
__device__ __forceinline__ void chain_st4(float* p, const float4 v) {
    if constexpr (SCOPED) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
        __hip_atomic_store(q, (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *reinterpret_cast<float4*>(p) = v;
    }
}
template <bool SCOPED>
__device__ __forceinline__ void chain_st1(float* p, float v) {
    if constexpr (SCOPED)
        __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

// One tile of one stage.  MODE 0: level blur + DoG; MODE 1: reduction (blur, kept pixels stored into the next octave).
template <int R, int MODE, bool SCOPED>
__device__ __forceinline__ void chain_tile(const ChainStage& st, int img, int t2, float* s_src, float* s_mid) {
    static_assert(R <= kChainMaxR, "LDS is sized for kChainMaxR");
    constexpr int TW = kChainTW, TH = kChainTH;
    constexpr int RA = chain_ra(R);
    constexpr int PAD = RA - R;
    constexpr int SWA = TW + 2 * RA;
    constexpr int SH = TH + 2 * R;
    constexpr int ROW4 = SWA / 4;
    constexpr int NL4 = SH * ROW4;
    constexpr int NLD = (NL4 + 255) / 256;
    constexpr int NT = 2 * R + 1;
    float4* s_src4 = reinterpret_cast<float4*>(s_src);
    float4* s_mid4 = reinterpret_cast<float4*>(s_mid);
    const int tid = threadIdx.x;
    const int w = st.w, h = st.h;
    const int ty = t2 / st.tiles_x, tx = t2 - ty * st.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const float* src = st.src + (size_t)img * (size_t)w * (size_t)h;
    // constant address space: the taps are never written while the kernel runs, which lets them live in SGPRs (scalar loads,
    // hoisted out of the passes) although the kernel stores through other pointers
    typedef const __attribute__((address_space(4))) float* ctaps_t;
    const ctaps_t taps = (ctaps_t)(uintptr_t)st.taps;

    // 1. source tile (+ halo, reflected at the image border) -> registers -> LDS; all loads are issued before the first is used
    {
        float4 pre[NLD];
        const bool interior = x0 - RA >= 0 && x0 + TW + RA <= w && y0 - R >= 0 && y0 + TH + R <= h;
        if (interior) {
            const float* base = src + (size_t)(y0 - R) * (size_t)w + (size_t)(x0 - RA);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = tid + 256 * i;
                if (e < NL4) {
                    const int ly = e / ROW4, c4 = e - ly * ROW4;
                    pre[i] = chain_ld4<SCOPED>(base + (size_t)ly * (size_t)w + (size_t)(4 * c4));
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = tid + 256 * i;
                if (e < NL4) {
                    const int ly = e / ROW4, c4 = e - ly * ROW4;
                    const size_t row = (size_t)chain_reflect(y0 - R + ly, h) * (size_t)w;
                    const int gx = x0 - RA + 4 * c4;
                    if (gx >= 0 && gx + 3 < w) {   // only the row is reflected (rows are 16-byte aligned: checked by the host)
                        pre[i] = chain_ld4<SCOPED>(src + row + (size_t)gx);
                    } else {
                        pre[i].x = chain_ld1<SCOPED>(src + row + (size_t)chain_reflect(gx + 0, w));
                        pre[i].y = chain_ld1<SCOPED>(src + row + (size_t)chain_reflect(gx + 1, w));
                        pre[i].z = chain_ld1<SCOPED>(src + row + (size_t)chain_reflect(gx + 2, w));
                        pre[i].w = chain_ld1<SCOPED>(src + row + (size_t)chain_reflect(gx + 3, w));
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + 256 * i;
            if (e < NL4) s_src4[e] = pre[i];
        }
    }
    lds_barrier();

    // 2. row pass: a wave reads 4 rows x 16 float4 per instruction, the rows RG apart (conflict-free lane groups)
    constexpr int RG = 16 / gcd_ce(ROW4, 16);
    constexpr int NBLK = (SH + 4 * RG - 1) / (4 * RG);
    const int rows_used = min(SH, h - y0 + 2 * R);   // the bottom tile row needs only the source rows its output rows reach
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
    if (y0 + rg * PY < h)
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
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // no hoisting of every LDS read to the top
        }
    // 4. stores
    const int x = x0 + 4 * cg;
    if (MODE == 0) {
        const size_t img_off = (size_t)img * (size_t)w * (size_t)h;
#pragma unroll
        for (int i = 0; i < PY; ++i) {
            const int y = y0 + rg * PY + i;
            if (y < h && x < w) {   // w is a multiple of 4: the four columns are inside together
                const size_t o = img_off + (size_t)y * (size_t)w + (size_t)x;
                const float4 prev = s_src4[(R + rg * PY + i) * ROW4 + (RA / 4) + cg];
                const float dx = acc[i].x - prev.x, dy = acc[i].y - prev.y;
                const float dz = acc[i].z - prev.z, dw = acc[i].w - prev.w;
                chain_st4<SCOPED>(st.dst + o, acc[i]);
                chain_st4<SCOPED>(st.dog + o, make_float4(128.0f + dx, 128.0f + dy, 128.0f + dz, 128.0f + dw));
            }
        }
    } else {
        int dcol[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dcol[j] = x + j < w ? st.inv_x[x + j] : -1;
        float* dst = st.dst + (size_t)img * (size_t)st.wd * (size_t)st.hd;
#pragma unroll
        for (int i = 0; i < PY; ++i) {
            const int y = y0 + rg * PY + i;
            if (y < h) {
                const int jd = st.inv_y[y];
                if (jd >= 0) {
                    const float av[4] = {acc[i].x, acc[i].y, acc[i].z, acc[i].w};
                    float* drow = dst + (size_t)jd * (size_t)st.wd;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (dcol[j] >= 0) chain_st1<SCOPED>(drow + dcol[j], av[j]);
                }
            }
        }
    }
}

// sync words, every counter on a 128-byte line of its own (512 workgroups hammering neighbouring words of ONE line made a draw
// take 6 - 10 us): [32 q] ticket of queue q, [256] error flag, [288 + 32 (stage * n + image)] tiles of (stage, image) completed
constexpr int kChainPad = 32;
constexpr int kChainErr = 8 * kChainPad;
constexpr int kChainSyncHead = 9 * kChainPad;

// SCOPED = false is the reference form: ordinary loads and stores with agent-scope release / acquire fences around every tile
// (option "chain_mode" = 0).  MIX (tests): every draw goes to the next queue, so the tiles of one image are spread over all XCDs.
template <bool SCOPED, bool MIX>
__global__ __launch_bounds__(256, 2) void blur_chain_kernel(ChainPlan cp, const unsigned* __restrict__ items, int* sync) {
    __shared__ __attribute__((aligned(16))) float s_src[kChainSrcFloats];
    __shared__ __attribute__((aligned(16))) float s_mid[kChainMidFloats];
    __shared__ int s_item;
    __shared__ int s_tickets[8];
    const int tid = threadIdx.x;
    int q = (int)(blockIdx.x & 7u);
    unsigned live = 0;
    for (int i = 0; i < 8; ++i)
        if (cp.q_off[i + 1] > cp.q_off[i]) live |= 1u << i;
    while (live) {
        if (!((live >> q) & 1u)) {
            q = (q + 1) & 7;
            continue;
        }
        const int qn = cp.q_off[q + 1] - cp.q_off[q];
        if (tid == 0) s_item = __hip_atomic_fetch_add(&sync[q * kChainPad], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int it = s_item;
        if (it >= qn) {   // this queue is empty: help the others - those that still have items (one look at all the tickets)
            live &= ~(1u << q);
            if (tid < 8) s_tickets[tid] = __hip_atomic_load(&sync[tid * kChainPad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();   // (also: s_item is rewritten by the next draw)
            for (int i = 0; i < 8; ++i)
                if (s_tickets[i] >= cp.q_off[i + 1] - cp.q_off[i]) live &= ~(1u << i);
            __syncthreads();
            continue;
        }
        const unsigned item = items[cp.q_off[q] + it];
        const int sidx = (int)(item >> 28), img = (int)((item >> 16) & 0xfffu), t2 = (int)(item & 0xffffu);
        const ChainStage& st = cp.st[sidx];
        if (st.dep >= 0) {
            if (tid == 0) {
                const int* flag = &sync[kChainSyncHead + (st.dep * cp.n_images + img) * kChainPad];
                int spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < st.dep_tiles) {
                    __builtin_amdgcn_s_sleep(4);
                    // ~0.1 s (a level takes microseconds): something is broken; leave instead of hanging, the host reports it
                    if (++spins > (1 << 17) || ((spins & 63) == 0 && __hip_atomic_load(&sync[kChainErr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        __hip_atomic_store(&sync[kChainErr], 1 + sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __syncthreads();
            if constexpr (!SCOPED) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every lane's loads come after the producers' stores
        }
        if (st.mode == 0) {
            switch (st.radius) {
                case 7: chain_tile<7, 0, SCOPED>(st, img, t2, s_src, s_mid); break;
                case 10: chain_tile<10, 0, SCOPED>(st, img, t2, s_src, s_mid); break;
                case 14: chain_tile<14, 0, SCOPED>(st, img, t2, s_src, s_mid); break;
                case 19: chain_tile<19, 0, SCOPED>(st, img, t2, s_src, s_mid); break;
                default: chain_tile<27, 0, SCOPED>(st, img, t2, s_src, s_mid); break;
            }
        } else {
            switch (st.radius) {
                case 10: chain_tile<10, 1, SCOPED>(st, img, t2, s_src, s_mid); break;
                default: chain_tile<14, 1, SCOPED>(st, img, t2, s_src, s_mid); break;
            }
        }
        // this lane's stores have reached the level at which the device is coherent: SCOPED, they are agent-scope accesses and
        // complete when the counter says so; otherwise the release fence writes the L2 back
        if constexpr (SCOPED)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();   // ... for every lane (also: the LDS tiles are free again)
        if (tid == 0) __hip_atomic_fetch_add(&sync[kChainSyncHead + (sidx * cp.n_images + img) * kChainPad], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Every single-lane block of this loop lies between two barriers of its own.  Without this one the compiler joined the
        // block above with the draw at the top of the next iteration and moved both out of the loop for lane 0, whose wave then
        // waited at the next barrier for a draw lane 0 could only make after the wave had left the loop (a launch that never ends).
        __syncthreads();
        if constexpr (MIX) q = (q + 1) & 7;
    }
}

__global__ void chain_touch_kernel() {}

}  // namespace

bool chain_radius_supported(int radius, int mode) {
    if (mode == 0) return radius == 7 || radius == 10 || radius == 14 || radius == 19 || radius == 27;
    return radius == 10 || radius == 14;
}

size_t chain_sync_ints(int n_stages, int n_images) { return (size_t)kChainSyncHead + (size_t)n_stages * (size_t)n_images * (size_t)kChainPad; }
int chain_sync_error_index() { return kChainErr; }

void launch_blur_chain(hipStream_t s, const ChainPlan& cp, const unsigned* d_items, int* d_sync, int mode, hipEvent_t ev_start, hipEvent_t ev_stop) {
    static thread_local int cap[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cap[dev] == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, blur_chain_kernel<true, false>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        (void)hipGetLastError();
        cap[dev] = per_cu * cus;
    }
    const int total = cp.q_off[8];
    int grid = total < cap[dev] ? total : cap[dev];
    if (grid >= 8) grid &= ~7;
    if (grid < 1) return;
    if (mode == 0)
        hipExtLaunchKernelGGL((blur_chain_kernel<false, false>), dim3((unsigned)grid), dim3(256), 0, s, ev_start, ev_stop, 0, cp, d_items, d_sync);
    else if (mode == 2)
        hipExtLaunchKernelGGL((blur_chain_kernel<true, true>), dim3((unsigned)grid), dim3(256), 0, s, ev_start, ev_stop, 0, cp, d_items, d_sync);
    else
        hipExtLaunchKernelGGL((blur_chain_kernel<true, false>), dim3((unsigned)grid), dim3(256), 0, s, ev_start, ev_stop, 0, cp, d_items, d_sync);
}

void tu_touch_chain(hipStream_t s) { hipLaunchKernelGGL(chain_touch_kernel, dim3(1), dim3(1), 0, s); }

}  // namespace sift_hip
