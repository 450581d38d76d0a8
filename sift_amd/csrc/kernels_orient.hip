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
#include <cstdlib>

#include "common.h"
#include "linalg3.h"
#include "grad_math.h"

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

// alg::orientationHistogram36's bin of a sample (algorithms.cpp:126-128): (u16)floor(orientation / 10) % 35.
// a is in [0, 360).  RN(a / 10) is monotone, so a < 9.5 gives floor 0 and 350.5 <= a < 360 gives floor 35, i.e.
// bin 0 after % 35 either way - which is every pixel, the reference feeding radians (App. B-9); the division
// only runs for values in between.
__device__ __forceinline__ unsigned orientation_bin(float a) {
    unsigned bin = 0u;
    if (!(a < 9.5f || (a >= 350.5f && a < 360.0f))) bin = f32_to_u16_x86(__builtin_floorf(a / 10.0f)) % 35u;
    return bin;
}

// One pixel of the gradient maps (sift.cpp:130-160 with alg::gradientMagnitude / Orientation, algorithms.cpp:108-116):
// magnitude, orientation, and the per-pixel inputs of alg::orientationHistogram36 (algorithms.cpp:126-128), which
// reads the INITIAL maps: weight = magnitude * gaussian, bin = (u16)floor(orientation / 10) % 35.
__device__ __forceinline__ void gradient_pixel(bool interior, float left, float right, float up, float down, float centre,
                                               float& m, float& a, float& pr, unsigned& bin) {
    m = 0.0f;
    a = 0.0f;
    if (interior) {   // border pixels stay 0
        const float dx = right - left;
        const float dy = down - up;
        // std::sqrt(std::pow(dx, 2) + std::pow(dy, 2)) evaluated in double (algorithms.cpp:109-110): grad_math.h
        m = gradient_magnitude(dx, dy);
        // std::fmod(atan2f(dy, dx) + 360.f, 360.) (algorithms.cpp:114-115); atan2f in [-pi, pi] so the
        // double fmod reduces to one conditional subtraction, which is exact.
        // branch-free common path (one division for every argument range); rare inputs take the full routine
        float r;
        if (!fdlibm_atan2f_common(dy, dx, r)) r = fdlibm_atan2f(dy, dx);
        // ... and needs no double: s = r + 360.f is a float in [360 - pi, 360 + pi]; below 360 fmod returns it unchanged, from 360
        // on s - 360 is a multiple of ulp(360) = 2^-15 below 4, i.e. a float, so the float subtraction is exact too and
        // (float)fmod((double)s, 360.) == s - 360.f bit for bit (a NaN stays the NaN it was)
        const float s = r + 360.0f;
        a = s >= 360.0f ? s - 360.0f : s;
    }
    pr = m * centre;
    bin = orientation_bin(a);
}

__global__ __launch_bounds__(256) void gradient_kernel(const float* __restrict__ g, float* __restrict__ mag,
                                                       float* __restrict__ ori, float* __restrict__ prod, int w, int h,
                                                       int* __restrict__ any_bin, int stamp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const size_t base = (size_t)blockIdx.z * (size_t)w * (size_t)h;
    const size_t o = base + (size_t)y * (size_t)w + (size_t)x;
    const bool interior = x >= 1 && x <= w - 2 && y >= 1 && y <= h - 2;
    float m, a, pr;
    unsigned bin;
    gradient_pixel(interior, interior ? g[o - 1] : 0.0f, interior ? g[o + 1] : 0.0f, interior ? g[o - (size_t)w] : 0.0f,
                   interior ? g[o + (size_t)w] : 0.0f, g[o], m, a, pr, bin);
    mag[o] = m;
    ori[o] = a;
    prod[o] = pr;
    // The reference feeds radians where degrees were meant (App. B-9): every sample lands in bin 0.  The
    // orientation stage skips the bin map of an image as long as this flag stays clear.
    if (bin != 0u && any_bin) any_bin[blockIdx.z] = stamp;   // (the batch's stamp, not 1: the flags are never cleared - launch_gradient)
}

// Four pixels of a row per thread (rows 16-byte aligned), kGradRows consecutive rows per thread: the three source rows a
// pixel needs roll through registers (row y+1 is fetched while row y is computed and becomes row y the step after), so the
// level is read 1 + 2 / kGradRows times instead of three times (consecutive workgroups are dealt to different XCDs, whose L2s
// do not share the neighbouring rows: measured 3.0x before).  16-byte loads and stores.  The samples' histogram bins are not
// stored (until round 4: one byte per pixel): every bin of an image is 0 unless its flag says otherwise, and the orientation
// stage then forms the bins from the orientation map itself.
constexpr int kGradRows = 16;

__global__ __launch_bounds__(256, 8) void gradient4_kernel(const float* __restrict__ g, float* __restrict__ mag,
                                                        float* __restrict__ ori, float* __restrict__ prod, int w, int h,
                                                        int* __restrict__ any_bin, int stamp) {
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y0 = blockIdx.y * kGradRows;
    if (x >= w) return;
    const size_t base = (size_t)blockIdx.z * (size_t)w * (size_t)h;
    const float* __restrict__ col = g + base + (size_t)x;
    const bool has_l = x >= 1, has_r = x + 4 <= w - 1;
    // Every load is issued with its row and column clamped into the image and nothing is zeroed afterwards: a clamped sample
    // only ever reaches a border pixel, whose results are discarded (gradient_pixel's `interior`).  No load sits under a branch
    // or behind a selected pointer (a select between the global row and a zero vector in private memory compiles to a FLAT load).
    const int lo_off = has_l ? -1 : 0, ro_off = has_r ? 4 : 3;
    auto rowc = [&](int y) { return (size_t)min(max(y, 0), h - 1) * (size_t)w; };
    auto row4 = [&](int y) { return *reinterpret_cast<const float4*>(col + rowc(y)); };
    auto left = [&](int y) { return col[(ptrdiff_t)rowc(y) + lo_off]; };
    auto right = [&](int y) { return col[(ptrdiff_t)rowc(y) + ro_off]; };
    float4 u4 = row4(y0 - 1), c4 = row4(y0);
    float lf = left(y0), rt = right(y0);
    unsigned any = 0u;
    const int y1 = min(y0 + kGradRows, h);
    for (int y = y0; y < y1; ++y) {
        const float4 d4 = row4(y + 1);
        const float nlf = left(y + 1), nrt = right(y + 1);
        const bool row_in = y >= 1 && y <= h - 2;
        const float cv[6] = {lf, c4.x, c4.y, c4.z, c4.w, rt};
        const float uv[4] = {u4.x, u4.y, u4.z, u4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
        float m[4], a[4], pr[4];
        unsigned bin[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool interior = row_in && x + e >= 1 && x + e <= w - 2;
            gradient_pixel(interior, cv[e], cv[e + 2], uv[e], dv[e], cv[e + 1], m[e], a[e], pr[e], bin[e]);
        }
        const size_t o = base + (size_t)y * (size_t)w + (size_t)x;
        // streamed out: nothing in this launch reads the maps again, and lines that are not left dirty in L2 need no write-back
        // when the kernel ends
        typedef float f4nt __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store((f4nt){m[0], m[1], m[2], m[3]}, reinterpret_cast<f4nt*>(mag + o));
        __builtin_nontemporal_store((f4nt){a[0], a[1], a[2], a[3]}, reinterpret_cast<f4nt*>(ori + o));
        __builtin_nontemporal_store((f4nt){pr[0], pr[1], pr[2], pr[3]}, reinterpret_cast<f4nt*>(prod + o));
        any |= bin[0] | bin[1] | bin[2] | bin[3];
        u4 = c4; c4 = d4; lf = nlf; rt = nrt;
    }
    if (any != 0u && any_bin) any_bin[blockIdx.z] = stamp;
}

// Two phases per workgroup of 128 keypoints:
//  1. histogram.  Each wave takes 32 keypoints, 4 at a time (LDS for three workgroups per CU): the 16x16 windows of orientation /
//     magnitude / Gaussian are fetched with row-coalesced loads and the 256 (bin, magnitude*gauss)
//     pairs of each keypoint are staged in LDS in the reference's summation order (x outer, y inner);
//     then LANE k runs keypoint k's ordered sum, so 8 dependent chains advance side by side.  All
//     256 samples of a keypoint normally share one bin (the reference feeds radians where degrees
//     were meant), which makes the sum a single register chain; a keypoint with mixed bins takes
//     the general path (ordered read-modify-write of its 36 bins).
//  2. peaks: one THREAD per keypoint runs the serial Sift::_findPeaks / vertexParabola logic on its
//     histogram column in LDS, so 128 dependent chains run side by side instead of one per wave.
constexpr int kOrientGroup = 128;
constexpr int kOrientSub = 4;       // keypoints staged per wave at a time
constexpr int kOrientCol = 20;      // staged samples of one window column (16 + pad: 2-way instead of 8-way write conflicts)
constexpr int kOrientStride = 324;  // floats (and bytes) between staged keypoints: lane k reads 16-byte words at bank 4k

__global__ __launch_bounds__(256) void orientation_kernel(const DevPlan* __restrict__ plan,
                                                          const OrientIn* __restrict__ oin,
                                                          const int* __restrict__ list_cnt, int list_cap,
                                                          OrientOut* __restrict__ out,
                                                          float* __restrict__ peaks_out, int* __restrict__ next_group,
                                                          const int* __restrict__ any_bin, int stamp, int dbg_arg) {
    const int dbg = dbg_arg & kDiagMask;   // measurement build only (common.h)
    // staging (phase 1) and the peak sets (phase 2) are never live together: they share storage
    __shared__ __attribute__((aligned(16))) float s_stage[4 * kOrientSub * kOrientStride];
    __shared__ __attribute__((aligned(16))) unsigned char s_sbin[4 * kOrientSub * kOrientStride];
    __shared__ float s_hist[36][kOrientGroup];
    __shared__ unsigned short s_kp[kOrientGroup];    // survivor-list position of each slot
    __shared__ unsigned char s_state[kOrientGroup];  // bit0 border-filtered, bits 1-2 throw code, bit7 run
    // what phase 1 needs to know about a keypoint's (octave, dog): looked up once per workgroup instead of a
    // chain of dependent scalar loads from the plan per keypoint
    struct LevelInfo {
        int wh;        // w | h << 16 of the nearest Gaussian level's octave
        int dead;      // dead_blur_radius code
        const float* prod;
        const float* ori;      // the samples' bins come from here (orientation_bin) when the image's flag is set
    };
    __shared__ LevelInfo s_lvl[kMaxLevels];
    static_assert(sizeof(float) * 4 * kOrientSub * kOrientStride >= sizeof(float) * 36 * kOrientGroup, "s_set overlay");
    float (*s_set)[kOrientGroup] = reinterpret_cast<float (*)[kOrientGroup]>(s_stage);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int img = blockIdx.y;
    const int cnt = list_cnt[img];
    const int D = plan->dogs;
    const size_t lbase = (size_t)img * (size_t)list_cap;
    float* __restrict__ wprod = s_stage + wv * kOrientSub * kOrientStride;
    unsigned char* __restrict__ wbin = s_sbin + wv * kOrientSub * kOrientStride;
    if (cnt <= 0) return;
    const bool bins_zero = any_bin != nullptr && any_bin[img] != stamp;   // block-uniform: every sample of this image has bin 0
    for (int l = tid; l < plan->octaves * D; l += 256) {
        const int lvl = plan->nearest_level[l];
        const int no = lvl / (D + 1);
        LevelInfo li;
        li.wh = plan->w[no] | (plan->h[no] << 16);
        li.dead = plan->dead_blur_radius[l];
        li.prod = plan->prod[lvl];
        li.ori = plan->ori[lvl];
        s_lvl[l] = li;
    }
    // The survivor count lives on the device; the groups of an image are dealt round-robin to its workgroups.  (Until round 3
    // they were drawn from a per-image counter: ~10 k atomics of 4096 workgroups on the ONE cache line that holds the 32
    // counters - 300 -> 284 us without them, tools/chain_trace.sh with FILTER=orient.)
    for (int grp = (int)blockIdx.x;; grp += (int)gridDim.x) {
        __syncthreads();
        if (grp * kOrientGroup >= cnt) break;
        // ---- phase 1 --------------------------------------------------------------------------
        // Bookkeeping of the wave's 32 keypoints with one keypoint per lane (records fetched coalesced, level looked up,
        // border and dead-blur tests, state and survivor position stored, address of the window's first pixel formed): what the
        // load loop below needs of keypoint i is then two or three lane reads, not a chain of scalar loads and tests per keypoint.
        const int slot_w = wv * (kOrientGroup / 4);          // first slot of this wave
        const int j_lane = grp * kOrientGroup + slot_w + lane;
        const bool have = lane < kOrientGroup / 4 && j_lane < cnt;
        unsigned long long org_prod = 0ull, org_bin = 0ull;  // byte address of the window's first sample in the weight / orientation maps
        int pitch = 0;                                       // row pitch of that level (pixels)
        bool run_l = false;
        if (have) {
            const OrientIn rec = oin[lbase + j_lane];
            const int x = rec.x, y = rec.y;
            const int l = (int)rec.octave * D + (int)rec.index;
            const LevelInfo li = s_lvl[l];
            const int w = li.wh & 0xffff, h = (int)((unsigned)li.wh >> 16);
            const bool border = (x < kRegion || x >= w - kRegion) || (y < kRegion || y >= h - kRegion);  // sift.cpp:173-174
            const int throws = border ? 0 : li.dead;  // sift.cpp:184 (0 ok, else error code)
            run_l = !border && throws == 0 && !(dbg & 4);
            s_kp[slot_w + lane] = (unsigned short)rec.kp;
            s_state[slot_w + lane] = (unsigned char)((border ? 1 : 0) | (throws << 1) | ((!border && throws == 0) ? 0x80 : 0));
            const size_t o = (size_t)img * (size_t)w * (size_t)h + (size_t)(y - kRegion) * (size_t)w + (size_t)(x - kRegion);
            org_prod = (unsigned long long)(uintptr_t)(li.prod + o);
            org_bin = (unsigned long long)(uintptr_t)(li.ori + o);
            pitch = w;
        }
        const unsigned runmask32 = (unsigned)__ballot(run_l);   // bit i: keypoint slot_w + i runs
        for (int sb = 0; sb < kOrientGroup / 4 / kOrientSub; ++sb) {
            const int slot0 = slot_w + sb * kOrientSub;
            if (grp * kOrientGroup + slot0 >= cnt) break;  // wave-uniform
            const unsigned runmask = (runmask32 >> (sb * kOrientSub)) & ((1u << kOrientSub) - 1u);   // wave-uniform bit per staged keypoint
            unsigned unimask = 0;
            // all window loads of the staged keypoints are issued before any of them is consumed
            float pp[kOrientSub][4];
            unsigned pb[kOrientSub][4];
#pragma unroll
            for (int k = 0; k < kOrientSub; ++k) {
#pragma unroll
                for (int it = 0; it < 4; ++it) { pp[k][it] = 0.0f; pb[k][it] = 0u; }
                if ((runmask >> k) & 1u) {  // wave-uniform
                    const int src = sb * kOrientSub + k;   // the lane that holds this keypoint's bookkeeping
                    const unsigned long long a64 =
                        (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)org_prod, src) |
                        ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(org_prod >> 32), src) << 32);
                    // (pointers made from integers are generic: spelled as global ones, or the loads become FLAT loads, which also
                    // count against the LDS counter the staging below waits on)
                    typedef const __attribute__((address_space(1))) float* gfloat_p;
                    const gfloat_p gp = (gfloat_p)(uintptr_t)a64;
                    const int w = __builtin_amdgcn_readlane(pitch, src);
                    if (bins_zero) {
                        // one 16-byte load per lane covers the whole window: lane = (row, 4-column group)
                        const size_t o = (size_t)(lane >> 2) * (size_t)w + (size_t)(4 * (lane & 3));
                        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                        typedef const __attribute__((address_space(1))) f4u* gf4u_p;
                        const f4u v4 = *(gf4u_p)(gp + o);
                        pp[k][0] = v4.x; pp[k][1] = v4.y; pp[k][2] = v4.z; pp[k][3] = v4.w;
                    } else {
                        const unsigned long long b64 =
                            (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)org_bin, src) |
                            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(org_bin >> 32), src) << 32);
                        const gfloat_p gb = (gfloat_p)(uintptr_t)b64;
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int ly = it * 4 + (lane >> 4);
                            const int lx = lane & 15;
                            const size_t o = (size_t)ly * (size_t)w + (size_t)lx;
                            pp[k][it] = gp[o];
                            pb[k][it] = orientation_bin(gb[o]);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < kOrientSub; ++k) {
                if (!((runmask >> k) & 1u)) continue;  // wave-uniform
                if (bins_zero) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int lx = 4 * (lane & 3) + it, ly = lane >> 2;
                        wprod[k * kOrientStride + lx * kOrientCol + ly] = pp[k][it];
                    }
                    if (lane == 0) wbin[k * kOrientStride] = 0;
                    unimask |= 1u << k;
                    continue;
                }
                unsigned first_bin = 0;
                bool same = true;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int ly = it * 4 + (lane >> 4);
                    const int lx = lane & 15;
                    const float sum = pp[k][it];
                    const unsigned i = pb[k][it];
                    wprod[k * kOrientStride + lx * kOrientCol + ly] = sum;
                    wbin[k * kOrientStride + lx * kOrientCol + ly] = (unsigned char)i;
                    if (it == 0) first_bin = __shfl(i, 0);
                    same = same && (i == first_bin);
                }
                if (__all(same)) unimask |= 1u << k;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // lane k: ordered sum of keypoint k (x outer, y inner == staging order)
            float acc = 0.0f;
            unsigned b0 = 0;
            if (!(dbg & 2) && lane < kOrientSub && ((runmask & unimask) >> lane) & 1u) {
                const float4* __restrict__ pv = reinterpret_cast<const float4*>(wprod + lane * kOrientStride);
                b0 = wbin[lane * kOrientStride];
#pragma unroll 4
                for (int cx = 0; cx < 16; ++cx) {   // x outer, y inner: column cx, rows 0..15
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 v = pv[cx * (kOrientCol / 4) + j];
                        acc += v.x;
                        acc += v.y;
                        acc += v.z;
                        acc += v.w;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < kOrientSub; ++k) {
                if (!((runmask >> k) & 1u)) continue;  // wave-uniform
                const int slot = slot0 + k;
                if ((unimask >> k) & 1u) {
                    const float a = __shfl(acc, k);
                    const unsigned b = __shfl(b0, k);
                    if (lane < 36) s_hist[lane][slot] = ((unsigned)lane == b) ? a : 0.0f;
                } else {
                    // mixed bins: bins[i] += sum sample by sample, in order (one lane, LDS resident)
                    if (lane < 36) s_hist[lane][slot] = 0.0f;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (lane == 0) {
                        for (int q = 0; q < 256; ++q) {
                            const int at = k * kOrientStride + (q >> 4) * kOrientCol + (q & 15);
                            const unsigned b = wbin[at];
                            s_hist[b][slot] = s_hist[b][slot] + wprod[at];
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        __syncthreads();
        // ---- phase 2 --------------------------------------------------------------------------
        if (tid < kOrientGroup && grp * kOrientGroup + tid < cnt) {
            const int kp = (int)s_kp[tid];
            const unsigned st = s_state[tid];
            OrientOut r;
            r.orientation = 0.0f;
            r.npeaks = 0;
            r.filtered = (unsigned char)(st & 1u);
            r.throws = (unsigned char)((st >> 1) & 3u);
            if ((st & 0x80u) && !(dbg & 1)) {
                // Sift::_findPeaks (sift.cpp:220-286) on histogram column `tid`: the 36 bins are pulled into
                // registers first (independent LDS reads in flight), the serial logic then runs on registers
                float hv[36];
#pragma unroll
                for (int i = 0; i < 36; ++i) hv[i] = s_hist[i][tid];
                int max_index = 0;
                float hmax = hv[0];
#pragma unroll
                for (int i = 1; i < 36; ++i) {
                    const float v = hv[i];
                    if (hmax < v) { hmax = v; max_index = i; }  // std::max_element: first largest
                }
                const float range = (float)((double)hmax * 0.8);
                auto vertex_at = [&](int i) {
                    unsigned short lnx, rnx;
                    float lny, rny;
                    const unsigned short px = (unsigned short)(i * 10 + 5);
                    const float py = s_hist[i][tid];
                    if (i == 0) { lnx = 355; lny = s_hist[35][tid]; }
                    else        { lnx = (unsigned short)((i - 1) * 10 + 5); lny = s_hist[i - 1][tid]; }
                    if (i == 35) { rnx = 5; rny = s_hist[0][tid]; }
                    else         { rnx = (unsigned short)((i + 1) * 10 + 5); rny = s_hist[i + 1][tid]; }
                    return vertex_parabola(lnx, lny, px, py, rnx, rny);
                };
                // peaks_only after the 80% threshold and the in-place local-maximum sweep (i = 1..34 uses the
                // already updated left neighbour and the not yet updated right): which bins survive, as a bit set
                unsigned long long peak_bits = 0ull;
                {
                    auto thresholded = [&](float v) { return (v < range) ? -1.0f : v; };
                    float left = thresholded(hv[0]);      // peaks_only[0], never touched by the sweep
                    float cur = thresholded(hv[1]);
                    if (left > -1.0f) peak_bits |= 1ull;
#pragma unroll
                    for (int i = 1; i < 36; ++i) {
                        const float right = (i < 35) ? thresholded(hv[i < 35 ? i + 1 : 35]) : 0.0f;
                        float only_i = cur;
                        if (i < 35 && (cur < left || cur < right)) only_i = -1.0f;
                        left = only_i;
                        cur = right;
                        if (only_i > -1.0f) peak_bits |= 1ull << i;
                    }
                    peak_bits &= ~(1ull << max_index);
                }
                // std::set<float>::emplace: a NaN first element blocks every later insert (no key
                // compares less than it and it compares less than none); a later NaN is never
                // inserted; numbers insert sorted and unique.
                const float v0 = vertex_at(max_index);
                s_set[0][tid] = v0;
                int n = 1;
                const bool nan_first = (v0 != v0);
                while (peak_bits) {   // ascending bin order, as the reference's loop
                    const int i = __ffsll((long long)peak_bits) - 1;
                    peak_bits &= peak_bits - 1ull;
                    const float v = vertex_at(i);
                    if (!nan_first && v == v) {
                        int pos = 0;
                        while (pos < n && s_set[pos][tid] < v) ++pos;
                        if (!(pos < n && !(v < s_set[pos][tid]))) {  // not equivalent to an existing key
                            for (int j = n; j > pos; --j) s_set[j][tid] = s_set[j - 1][tid];
                            s_set[pos][tid] = v;
                            ++n;
                        }
                    }
                }
                r.orientation = s_set[0][tid];
                r.npeaks = (unsigned short)n;
                if (n > 1) {
                    float* po = peaks_out + (lbase + (size_t)kp) * 36;
                    for (int j = 0; j < n; ++j) po[j] = s_set[j][tid];
                }
            }
            out[lbase + (size_t)kp] = r;
        }
        __syncthreads();
    }
}

// d_any_bin[image] is set to `stamp` when some sample of the image has a histogram bin other than 0 (never, with the reference's
// radians: App. B-9).  The stamp is the context's batch number, so the flags need no clearing between batches (round 5: the
// one-workgroup clearing launch in front of the gradient pass waited ~200 us for a slot beside the extrema pass, and the
// gradient pass behind it); a stale word that happens to equal the stamp only sends the image down the general path, which is
// always right.
void launch_gradient(hipStream_t s, const float* g, float* mag, float* ori, float* prod, int w, int h,
                     int n, int* d_any_bin, int stamp, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const bool vec = (w & 3) == 0 && ((((uintptr_t)g | (uintptr_t)mag | (uintptr_t)ori | (uintptr_t)prod) & 15u) == 0);
    if (vec) {
        const dim3 grid4((unsigned)((w / 4 + 255) / 256), (unsigned)((h + kGradRows - 1) / kGradRows), (unsigned)n);
        hipExtLaunchKernelGGL(gradient4_kernel, grid4, dim3(256), 0, s, ev_start, ev_stop, 0, g, mag, ori, prod, w, h, d_any_bin, stamp);
        return;
    }
    const dim3 grid((unsigned)((w + 255) / 256), (unsigned)h, (unsigned)n);
    hipExtLaunchKernelGGL(gradient_kernel, grid, dim3(256), 0, s, ev_start, ev_stop, 0, g, mag, ori, prod, w, h, d_any_bin, stamp);
}

static int g_orient_dbg = 0;
void set_orient_dbg(int v) { g_orient_dbg = v; }

void launch_orientation(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan, const Candidate* d_cands,
                        const OrientIn* d_oin, const int* d_list_cnt, int list_cap, OrientOut* d_out,
                        float* d_peaks, int* d_next_group, const int* d_any_bin, int zero_counters, int stamp) {
    (void)d_cands;
    const dim3 grid(128, (unsigned)plan.n_images);
    const int dbg = g_orient_dbg;   // option "orient_dbg": timing ablations only
    // zero_counters: how many consecutive per-image counter arrays to clear first (0: the caller already did)
    (void)zero_counters;   // the groups are dealt statically: no counters to clear
    hipLaunchKernelGGL(orientation_kernel, grid, dim3(256), 0, s, d_plan, d_oin, d_list_cnt, list_cap, d_out,
                       d_peaks, d_next_group, d_any_bin, stamp, dbg);
}

// The runtime builds a translation unit's device code on the first launch of any of its kernels, and two host threads that make
// their first launches at the same time (several contexts, one thread each) were seen to crash inside that step
// (tools/asan_example.sh: SEGV below hipLaunchKernel).  sift_hip_create touches every unit once, under a lock.
__global__ void tu_probe_orient_kernel() {}
void tu_touch_orient(hipStream_t s) { hipLaunchKernelGGL(tu_probe_orient_kernel, dim3(1), dim3(1), 0, s); }

}  // namespace sift_hip
