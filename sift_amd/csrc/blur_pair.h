// The kernels that form the pyramid's first two levels in one launch (sift_amd/csrc/kernels_pair.hip has the story and the
// launchers; tools/probe/blur_probe.hip times them alone).  Device code only; needs <hip/hip_runtime.h>.
#pragma once

#pragma clang fp contract(off)

namespace sift_hip {

__device__ __forceinline__ int pair_reflect(int p, int n) {
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * (n - 1) - p : p;
    return p < 0 ? 0 : (p >= n ? n - 1 : p);
}

constexpr int kPairPF = 4;
constexpr int pair_runin(int r) { return 2 * r + (kPairPF - (3 * r) % kPairPF) % kPairPF; }   // 2R + E with 2R + E + R a multiple of PF
constexpr int pair_ra(int r) { return (r + 3) / 4 * 4; }
constexpr int pair_useful(int r) { return 256 - 2 * pair_ra(r); }

template <int R>
__global__ __launch_bounds__(64) void blur_pair_kernel(const float* __restrict__ in, float* __restrict__ g0,
                                                        float* __restrict__ g1, int w, int h, int strips, int chunks,
                                                        int chunk_h, int total_units, const float* __restrict__ taps) {
    constexpr int CPL = 4;
    constexpr int PF = kPairPF;
    constexpr int RI = pair_runin(R);   // stage 1's run-in, padded so that stage 2 starts on a whole unrolled body
    constexpr int E = RI - 2 * R;
    constexpr int RA = pair_ra(R);
    constexpr int PAD = RA - R;
    constexpr int HX = 2 * R - RA;   // input columns beyond the 256 of the strip that stage 1 needs, each side
    static_assert(HX >= 0 && HX <= RA, "strip geometry");
    constexpr int NFIX = RA + HX;    // columns filled by reflection where a strip touches the image's edge
    constexpr int NT = 2 * R + 1;
    constexpr int ROWF = 64 * CPL + 2 * RA;
    constexpr int SWU = pair_useful(R);
    constexpr int DP2 = 2 * R + 1;   // rows of g(0,0) live in the ring: stage 2 reads at most 2R rows behind the row stage 1 just wrote
    constexpr int LAG = R;           // stage 2 starts R rows of g(0,0) behind stage 1
    constexpr int NV = PAD + CPL + 2 * R;
    constexpr int NV4 = (NV + CPL - 1) / CPL;
    constexpr int EDGE_LANES = RA / CPL;  // lanes each side whose columns are halo only
    typedef float f4v __attribute__((ext_vector_type(CPL)));
    // one wave per workgroup: 12 rows of 272 floats (R = 5) = 13 KB, twelve workgroups per CU = the three waves per SIMD the
    // registers allow
    __shared__ __attribute__((aligned(16))) float s_src[ROWF];
    __shared__ __attribute__((aligned(16))) float s_ring[DP2 * ROWF];

    const int lane = threadIdx.x & 63;
    const int unit = (int)blockIdx.x;
    if (unit >= total_units) return;
    const int per_img = strips * chunks;
    const int img = unit / per_img;
    const int rem = unit - img * per_img;
    const int chunk = rem / strips;
    const int strip = rem - chunk * strips;
    const int xs = min(strip * SWU, w - SWU);   // the last strip is pulled left to end at the image's last column
    const int x0 = xs - RA;                     // column of lane 0
    // columns of the wave's row (x0 - HX .. x0 + 256 + HX) that lie outside the image are filled by reflection
    const bool left_edge = x0 - HX < 0, right_edge = x0 + 64 * CPL + HX > w;
    const int c0 = RA - x0, c1 = RA + (w - 1 - x0);   // LDS index of column 0 / of column w - 1
    const bool fix_l = lane < NFIX && c0 - (lane + 1) >= RA - HX;
    const bool fix_r = lane < NFIX && c1 + lane + 1 <= RA + 64 * CPL + HX - 1;
    const int y0 = min(chunk * chunk_h, h - chunk_h);
    const int y1 = y0 + chunk_h;
    const int ra = max(0, y0 - R), rb = min(h - 1, y1 - 1 + R);   // rows of g(0,0) this wave forms
    const int n1 = rb - ra + 1 + RI;                              // stage 1 steps
    const int n2 = chunk_h + 2 * R;                               // stage 2 steps
    const int s2 = RI + LAG;                                      // step at which stage 2 takes its first row
    const int nsteps = (s2 + n2 + PF - 1) / PF * PF;
    const int p0 = ra - R - E;

    const size_t img_off = (size_t)img * (size_t)w * (size_t)h;
    const float* __restrict__ src = in + img_off;
    float* srow = s_src;
    float* ring = s_ring;

    const int mcol = min(max(x0 + CPL * lane, 0), w - CPL);   // lanes outside the image load a legal address; LDS fixes the value
    const unsigned moff = 4u * (unsigned)mcol;
    const bool has_halo = lane < 2 * HX;
    const int hl = has_halo ? lane : 0;
    const int hcol = pair_reflect(hl < HX ? x0 - HX + hl : x0 + 64 * CPL + (hl - HX), w);
    const int hslot = hl < HX ? RA - HX + hl : RA + 64 * CPL + (hl - HX);
    const unsigned hoff = 4u * (unsigned)hcol;
    const bool useful = lane >= EDGE_LANES && lane < 64 - EDGE_LANES;
    const unsigned ooff = 4u * (unsigned)(x0 + CPL * lane);   // (only used by the useful lanes)

    float tp[NT];   // both blurs use the same taps (gauss_scale[1] == sigma, sift.cpp:388-398)
#pragma unroll
    for (int k = 0; k < NT; ++k) tp[k] = taps[k];
    f4v A1[NT], A2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) A1[j] = A2[j] = (f4v)(0.0f);

    f4v pm[PF];
    float ph[PF];
#define SIFT_PAIR_FETCH(T, U)                                                                                     \
    {                                                                                                             \
        const char* rowp_ = reinterpret_cast<const char*>(src + (size_t)pair_reflect(p0 + (T), h) * (size_t)w);   \
        pm[U] = *reinterpret_cast<const f4v*>(rowp_ + moff);                                                      \
        ph[U] = *reinterpret_cast<const float*>(rowp_ + hoff);                                                    \
    }
    // columns outside the image, filled from their mirror images inside it (both within the wave's row)
#define SIFT_PAIR_EDGES(ROW)                                                                                      \
    {                                                                                                             \
        if (left_edge || right_edge) {                                                                            \
            __builtin_amdgcn_wave_barrier();                                                                      \
            if (left_edge && fix_l) (ROW)[c0 - 1 - lane] = (ROW)[c0 + 1 + lane];                                  \
            if (right_edge && fix_r) (ROW)[c1 + 1 + lane] = (ROW)[c1 - 1 - lane];                                 \
        }                                                                                                         \
        __builtin_amdgcn_wave_barrier();                                                                          \
    }
    // row pass of the window at ROW (lane's 4 outputs), then the column sums slide by one row
#define SIFT_PAIR_BLUR(ROW, TP, A)                                                                                \
    {                                                                                                             \
        f4v W_[NV4];                                                                                              \
        const f4v* p4_ = reinterpret_cast<const f4v*>(ROW) + lane;                                                \
        _Pragma("unroll") for (int c = 0; c < NV4; ++c) W_[c] = p4_[c];                                           \
        float v_[NV4 * CPL];                                                                                      \
        _Pragma("unroll") for (int c = 0; c < NV4; ++c)                                                           \
            _Pragma("unroll") for (int e = 0; e < CPL; ++e) v_[CPL * c + e] = W_[c][e];                           \
        f4v m_ = (f4v)(0.0f);                                                                                     \
        _Pragma("unroll") for (int k = 0; k < NT; ++k) {                                                          \
            const float tap_ = TP[NT - 1 - k];                                                                    \
            _Pragma("unroll") for (int e = 0; e < CPL; ++e) m_[e] += tap_ * v_[PAD + k + e];                      \
        }                                                                                                         \
        f4v An_[NT];                                                                                              \
        _Pragma("unroll") for (int i = 0; i <= R; ++i) {                                                          \
            const f4v pr_ = TP[i] * m_;   /* taps are symmetric bit for bit: one product serves slots i and 2R-i */ \
            An_[i] = A[i + 1] + pr_;                                                                              \
            if (2 * R - i != i) An_[2 * R - i] = (2 * R - i + 1 < NT ? A[2 * R - i + 1] : (f4v)(0.0f)) + pr_;     \
        }                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) A[j] = An_[j];                                             \
    }

#pragma unroll
    for (int u = 0; u < PF; ++u) SIFT_PAIR_FETCH(u, u)

    // one step of each stage (S: step, U = S mod PF: the prefetch register of its input row)
#define SIFT_PAIR_STAGE1(S, U)                                                                                    \
    {                                                                                                             \
        *reinterpret_cast<f4v*>(srow + RA + CPL * lane) = pm[U];                                                  \
        if (has_halo) srow[hslot] = ph[U];                                                                        \
        SIFT_PAIR_EDGES(srow)                                                                                     \
        SIFT_PAIR_FETCH((S) + PF, U)                                                                              \
        SIFT_PAIR_BLUR(srow, tp, A1)                                                                              \
        if ((S) >= RI) {                                                                                          \
            const int r_ = ra + (S) - RI;   /* the row of g(0,0) this step completed */                           \
            float* rrow_ = ring + (unsigned)r_ % (unsigned)DP2 * ROWF;                                            \
            *reinterpret_cast<f4v*>(rrow_ + RA + CPL * lane) = A1[0];                                             \
            SIFT_PAIR_EDGES(rrow_)                                                                                \
            if (r_ >= y0 && r_ < y1 && useful)                                                                    \
                __builtin_nontemporal_store(A1[0], reinterpret_cast<f4v*>(reinterpret_cast<char*>(g0 + img_off + (size_t)r_ * (size_t)w) + ooff)); \
        }                                                                                                         \
    }
#define SIFT_PAIR_STAGE2(S)                                                                                       \
    {                                                                                                             \
        const int e2_ = (S) - s2;   /* stage 2's stream index */                                                  \
        const int q_ = pair_reflect(y0 - R + e2_, h);                                                             \
        const float* qrow_ = ring + (unsigned)q_ % (unsigned)DP2 * ROWF;                                          \
        SIFT_PAIR_BLUR(qrow_, tp, A2)                                                                             \
        if (e2_ >= 2 * R && e2_ < n2 && useful)                                                                   \
            __builtin_nontemporal_store(A2[0], reinterpret_cast<f4v*>(reinterpret_cast<char*>(g1 + img_off + (size_t)(y0 + e2_ - 2 * R) * (size_t)w) + ooff)); \
        __builtin_amdgcn_wave_barrier();   /* the ring row is read before a later step rewrites it */             \
    }

    // Three loops, so that neither stage sits under a branch in the long one (a branch around a stage makes the compiler
    // keep the sliding sums in fixed registers: 44 moves per stage and step): stage 1 alone until stage 2's first row is
    // there, both, then what is left of either.
    int s0 = 0;
#pragma unroll 1
    for (; s0 < s2; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            SIFT_PAIR_STAGE1(s0 + u, u)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int both_end = s2 + (min(n1, s2 + n2) - s2) / PF * PF;
#pragma unroll 1
    for (; s0 < both_end; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            SIFT_PAIR_STAGE1(s0 + u, u)
            __builtin_amdgcn_sched_barrier(0);   // (the two stages interleaved cost more registers than three waves per SIMD leave)
            SIFT_PAIR_STAGE2(s0 + u)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll 1
    for (; s0 < nsteps; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (s0 + u < n1) SIFT_PAIR_STAGE1(s0 + u, u)
            if (s0 + u < s2 + n2) SIFT_PAIR_STAGE2(s0 + u)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef SIFT_PAIR_STAGE2
#undef SIFT_PAIR_STAGE1
#undef SIFT_PAIR_BLUR
#undef SIFT_PAIR_EDGES
#undef SIFT_PAIR_FETCH
}


// (Round 6 also built this launch with ROWS packed in pairs - blur_stream.h: blur_stream2_kernel has the idea -, 128 columns per
// wave, the second blur's row pass run once per row of g(0,0) with its results kept in a per-lane ring in LDS: bit-identical,
// 150 VGPRs, three waves per SIMD, and SLOWER alone: 261 us at its best cut (2720 waves) against 240 us for the kernel above in
// the same probe (profiles/r06_pair_probe.txt).  With two columns per lane the per-row bookkeeping of TWO stages - guarded stores
// of two levels, the edge fills of two lines, ring slots, reflected row indices - is spread over half as many columns as here:
// 87 instructions per row and lane column against 81.  Removed again; what the launch above leaves on the table is in DESIGN.md.)
}  // namespace sift_hip
