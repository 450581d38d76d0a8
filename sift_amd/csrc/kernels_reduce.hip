// alg::reduceToNextLevel (/root/reference/algorithms.cpp:24-36): Gaussian blur of a level followed by
// resizeImageNoInterpolation to ((w+1)/2, (h+1)/2), as ONE pass that evaluates only the pixels the resampling keeps.
//
// The resampling keeps source pixel (lutx[i], luty[j]) for destination (i, j), and its index map (Vigra's accumulated-double
// rule, host_glue.cpp resize_index_map) is  lut[i] = 2 i + p(i)  with p = 0 up to a split point and p = 1 from there on (no
// split at all for an odd source size).  The reference's two passes are
//     tmp(x, y) = sum_k tap[2R-k] * src(x-R+k, y)        (every x of a KEPT column, every y)
//     out(x, y) = sum_k tap[2R-k] * tmp(x, y-R+k)        (KEPT x, KEPT y)
// so a kept pixel needs the row pass at its own column only, on every source row, and the column pass on kept rows only:
// (2R+1) + (3R+2)/4 multiply-adds per SOURCE pixel instead of 2(2R+1) + (3R+2) for the full-resolution blur that
// blur_stream_kernel<..., DEC> (kernels_pyramid.hip) computes before it drops three quarters of it.
//
// Work unit = one wave: a strip of up to 64 * DCPL destination columns with one column parity p, a chunk of destination rows
// with one row parity, walked top to bottom over the SOURCE rows:
//   * a lane fetches the 2 * DCPL consecutive source columns of its DCPL destination columns (PF rows ahead, into registers;
//     the row base is wave-uniform) and drops them DE-INTERLEAVED into the wave's LDS row: columns of the kept parity into
//     S, the others into T, each as one 16-byte (8-byte) store.  With p = 1 the whole strip simply starts one source column
//     later (the loads are then only 4-byte aligned), so S always holds the kept parity;
//   * row pass: the window  x-R .. x+R  of a destination column alternates between S and T; a lane's windows for its DCPL
//     outputs are two aligned runs of S and T (16-byte LDS reads at a lane stride of 16 bytes: conflict-free), and the
//     sum runs in the reference's ascending order;
//   * column pass in registers, on PAIRS of source rows: output row u (centre at stream row R + 2u) receives the tap
//     d = t - R - 2u from stream row t, so even stream rows feed the R+1 open outputs with the even-offset taps and odd rows
//     feed R of them with the odd-offset taps; the R+1 partial sums slide by one output per pair,
//         C[q] = B[q] + tap_even[q] * m_even   (q = 0..R; B[0] = 0: a fresh output; C[R] is complete and stored)
//         B[q+1] = C[q] + tap_odd[q] * m_odd   (q = 0..R-1)
//     which is each output's ascending-order sum from 0.0f.  The taps are symmetric bit for bit, so one product serves the
//     two slots that use the same tap (as in blur_stream_kernel).
// No workgroup barrier; HBM sees each source row once per chunk plus 2R rows of run-in.
// Bit-exactness contract as kernels_pyramid.hip: no FMA, sums from 0.0f in ascending source order, reflect(p) = -p / 2(n-1)-p.
#include <hip/hip_ext.h>

#include "common.h"

#pragma clang fp contract(off)

namespace sift_hip {

namespace {

__device__ __forceinline__ int rd_reflect_clamp(int p, int n) {
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * (n - 1) - p : p;
    return p < 0 ? 0 : (p >= n ? n - 1 : p);
}

constexpr int rd_floor_half(int k) { return k >= 0 ? k / 2 : -((-k + 1) / 2); }
constexpr int rd_round_up(int v, int m) { return (v + m - 1) / m * m; }

constexpr int kReducePF = 4;   // source rows in flight per wave (two pairs)

// registers: (R+1) partial sums + two windows of DCPL + 2A floats + PF rows of 2 DCPL floats, per lane
constexpr int reduce_occ(int r, int dcpl) { return dcpl == 4 ? (r <= 8 ? 4 : r <= 12 ? 3 : 2) : (r <= 12 ? 4 : 3); }

}  // namespace

// Geometry of one launch: destination columns [0, sx) have parity 0 and are cut into nxa strips of swa columns, [sx, wd) have
// parity 1: nxb strips of swb (every strip a whole number of lane vectors); rows likewise with chunks of cha / chb rows.
struct ReduceGeom {
    int sx, nxa, swa, nxb, swb;
    int sy, nya, cha, nyb, chb;
};

template <int R, int DCPL>
__global__ __launch_bounds__(256, reduce_occ(R, DCPL)) void blur_reduce_kernel(const float* __restrict__ in, float* __restrict__ out, int w, int h,
                                                                               int wd, int hd, int total_units, ReduceGeom g,
                                                                               const float* __restrict__ taps) {
    constexpr int PF = kReducePF;
    constexpr int NT = 2 * R + 1;
    constexpr int CE = (R + 1) / 2;                 // odd offsets on one side
    constexpr int A = rd_round_up(CE, DCPL);         // halo slots each side of S and T, a whole number of lane vectors
    constexpr int ROWF = 64 * DCPL + 2 * A;          // floats per LDS row (S or T)
    constexpr int NW = (DCPL + 2 * A) / DCPL;        // lane vectors per window
    typedef float vec __attribute__((ext_vector_type(DCPL)));
    typedef float vec_u __attribute__((ext_vector_type(DCPL), aligned(4)));   // global accesses that may be only 4-byte aligned
    __shared__ __attribute__((aligned(16))) float s_row[4][2 * ROWF];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int unit = (int)blockIdx.x * 4 + wave;
    if (unit >= total_units) return;
    const int nx = g.nxa + g.nxb, ny = g.nya + g.nyb;
    const int img = unit / (nx * ny);
    const int rem = unit - img * (nx * ny);
    const int cy = rem / nx;
    const int cx = rem - cy * nx;
    // strip: destination columns [i0, i0 + cnt), source column of destination i is 2 i + px
    int i0, cnt, px;
    if (cx < g.nxa) { i0 = cx * g.swa; cnt = min(g.swa, g.sx - i0); px = 0; }
    else { i0 = g.sx + (cx - g.nxa) * g.swb; cnt = min(g.swb, wd - i0); px = 1; }
    // chunk: destination rows [j0, j0 + rows), source row of destination j is 2 j + py
    int j0, rows, py;
    if (cy < g.nya) { j0 = cy * g.cha; rows = min(g.cha, g.sy - j0); py = 0; }
    else { j0 = g.sy + (cy - g.nya) * g.chb; rows = min(g.chb, hd - j0); py = 1; }

    const float* __restrict__ src = in + (size_t)img * (size_t)w * (size_t)h;
    float* __restrict__ dst = out + (size_t)img * (size_t)wd * (size_t)hd;
    float* S = s_row[wave];
    float* T = S + ROWF;

    // Lanes beyond the strip shadow its last lane (same addresses, same values), so that every lane runs the same instruction
    // stream and no global access sits under a branch (the compiler's vmcnt bookkeeping stays exact, the prefetch in flight).
    const int nl = cnt / DCPL;
    const int el = min(lane, nl - 1);
    const int cb = 2 * i0 + px;                       // source column of relative column 0
    const int mcol = cb + 2 * DCPL * el;              // first of the lane's 2 * DCPL source columns
    // The last source column of the image's last strip can be column w (one past the row: w even, parity 1); the row pass
    // reads it as the reflection of column w, which is column w - 2: this lane's own last-but-two element.
    const bool fix_last = mcol + 2 * DCPL - 1 >= w;
    const unsigned moff = 4u * (unsigned)mcol;
    // halo: relative columns -R .. -1 and 2 cnt .. 2 cnt + R - 2, one per lane, reflected at the image border
    constexpr int NH = 2 * R - 1;
    const bool has_halo = lane < NH;
    const int hl = has_halo ? lane : 0;
    const int hrel = hl < R ? hl - R : 2 * cnt + (hl - R);
    const int hcol = rd_reflect_clamp(cb + hrel, w);
    const unsigned hoff = 4u * (unsigned)hcol;
    // relative column c sits in S[A + c/2] when c is even, in T[A + (c-1)/2] when odd (floor division)
    const int hfl = hrel >= 0 ? hrel >> 1 : -((-hrel + 1) >> 1);
    float* hslot = ((hrel & 1) ? T : S) + A + hfl;

    float tp[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) tp[k] = taps[k];

    vec B[R + 1];
#pragma unroll
    for (int q = 0; q <= R; ++q) B[q] = (vec)(0.0f);

    const int p0 = 2 * j0 + py - R;                   // source row of stream row 0
    vec pa[PF], pb[PF];                               // the lane's 2 * DCPL source columns of a row: first / second half
    float ph[PF];
#define SIFT_RD_FETCH(TT, U)                                                                                          \
    {                                                                                                                 \
        const char* rowp_ = reinterpret_cast<const char*>(src + (size_t)rd_reflect_clamp(p0 + (TT), h) * (size_t)w);  \
        pa[U] = *reinterpret_cast<const vec_u*>(rowp_ + moff);                                                        \
        pb[U] = *reinterpret_cast<const vec_u*>(rowp_ + moff + 4u * DCPL);                                            \
        ph[U] = *reinterpret_cast<const float*>(rowp_ + hoff);                                                        \
    }
    // stream row held in pa/pb/ph[U] -> LDS (de-interleaved), refill the slot from HBM, row pass -> m
#define SIFT_RD_ROW(TT, U, M)                                                                                         \
    {                                                                                                                 \
        vec se_, so_;                                                                                                 \
        if constexpr (DCPL == 4) {                                                                                    \
            se_ = vec{pa[U][0], pa[U][2], pb[U][0], pb[U][2]};                                                        \
            so_ = vec{pa[U][1], pa[U][3], pb[U][1], fix_last ? pb[U][1] : pb[U][3]};                                  \
        } else {                                                                                                      \
            se_ = vec{pa[U][0], pb[U][0]};                                                                            \
            so_ = vec{pa[U][1], fix_last ? pa[U][1] : pb[U][1]};                                                      \
        }                                                                                                             \
        __builtin_amdgcn_wave_barrier();   /* the previous row's window reads are done */                             \
        *reinterpret_cast<vec*>(S + A + DCPL * el) = se_;                                                             \
        *reinterpret_cast<vec*>(T + A + DCPL * el) = so_;                                                             \
        if (has_halo) *hslot = ph[U];                                                                                 \
        __builtin_amdgcn_wave_barrier();                                                                              \
        SIFT_RD_FETCH((TT) + PF, U)                                                                                   \
        vec ws_[NW], wt_[NW];                                                                                         \
        {                                                                                                             \
            const vec* ps_ = reinterpret_cast<const vec*>(S) + el;                                                    \
            const vec* pt_ = reinterpret_cast<const vec*>(T) + el;                                                    \
            _Pragma("unroll") for (int c = 0; c < NW; ++c) { ws_[c] = ps_[c]; wt_[c] = pt_[c]; }                      \
        }                                                                                                             \
        float vs_[NW * DCPL], vt_[NW * DCPL];                                                                         \
        _Pragma("unroll") for (int c = 0; c < NW; ++c)                                                                \
            _Pragma("unroll") for (int e = 0; e < DCPL; ++e) { vs_[DCPL * c + e] = ws_[c][e]; vt_[DCPL * c + e] = wt_[c][e]; } \
        M = (vec)(0.0f);                                                                                              \
        _Pragma("unroll") for (int k = -R; k <= R; ++k) {                                                             \
            const float tap = tp[R - k];                                                                              \
            _Pragma("unroll") for (int e = 0; e < DCPL; ++e)                                                          \
                M[e] += tap * ((k & 1) ? vt_[A + e + rd_floor_half(k - 1)] : vs_[A + e + rd_floor_half(k)]);          \
        }                                                                                                             \
    }
    // one pair of stream rows 2V, 2V+1 (held in slots U0, U0+1)
#define SIFT_RD_PAIR(V, U0)                                                                                           \
    {                                                                                                                 \
        vec me_, mo_;                                                                                                 \
        SIFT_RD_ROW(2 * (V), U0, me_)                                                                                 \
        vec C[R + 1];                                                                                                 \
        _Pragma("unroll") for (int q = 0; q <= R / 2; ++q) {                                                          \
            const vec pr = tp[2 * R - 2 * q] * me_;         /* the tap of slots q and R - q */                        \
            C[q] = B[q] + pr;                                                                                         \
            if (R - q != q) C[R - q] = B[R - q] + pr;                                                                 \
        }                                                                                                             \
        {                                                                                                             \
            const int u_ = (V) - R;                           /* the output C[R] completes: wave-uniform */           \
            if (u_ >= 0 && u_ < rows)                                                                                 \
                *reinterpret_cast<vec_u*>(dst + (size_t)(j0 + u_) * (size_t)wd + (size_t)(i0 + DCPL * el)) = C[R];   \
        }                                                                                                             \
        SIFT_RD_ROW(2 * (V) + 1, (U0) + 1, mo_)                                                                       \
        _Pragma("unroll") for (int q = 0; q < (R + 1) / 2; ++q) {                                                     \
            const vec pr = tp[2 * R - 2 * q - 1] * mo_;     /* the tap of slots q and R - 1 - q */                    \
            B[q + 1] = C[q] + pr;                                                                                     \
            if (R - 1 - q != q) B[R - q] = C[R - 1 - q] + pr;                                                         \
        }                                                                                                             \
        B[0] = (vec)(0.0f);                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }

#pragma unroll
    for (int u = 0; u < PF; ++u) SIFT_RD_FETCH(u, u)

    // outputs 0 .. rows-1 complete in pairs R .. R + rows - 1; two pairs per trip
    const int npairs = rows + R;
#pragma unroll 1
    for (int v = 0; v < npairs; v += 2) {
        SIFT_RD_PAIR(v, 0)
        SIFT_RD_PAIR(v + 1, 2)
    }
#undef SIFT_RD_PAIR
#undef SIFT_RD_ROW
#undef SIFT_RD_FETCH
}

static thread_local hipEvent_t t_rd_start = nullptr, t_rd_stop = nullptr;

template <int R, int DCPL>
static bool launch_reduce_rd(hipStream_t s, const float* in, float* dst, int w, int h, int wd, int hd, int n, const float* d_taps,
                             int sx, int sy, int min_waves) {
    // every strip a whole number of lane vectors; both parities' column counts must allow it
    if (sx % DCPL != 0 || (wd - sx) % DCPL != 0) return false;
    if (w < R + 1 || h < R + 1) return false;
    ReduceGeom g{};
    g.sx = sx; g.sy = sy;
    constexpr int SW = 64 * DCPL;
    auto cut = [](int len, int maxw, int quant, int& count, int& width) {
        count = len > 0 ? (len + maxw - 1) / maxw : 0;
        width = count ? ((len + count - 1) / count + quant - 1) / quant * quant : quant;
        if (count) count = (len + width - 1) / width;
    };
    cut(sx, SW, DCPL, g.nxa, g.swa);
    cut(wd - sx, SW, DCPL, g.nxb, g.swb);
    const int nx = g.nxa + g.nxb;
    // rows per chunk: ~2048 waves per launch (8 per CU), but at least 2R output rows a chunk (the 2R run-in source rows then cost a third)
    const int target = 2048;
    int want_chunks = target / (n * nx);
    if (want_chunks < 1) want_chunks = 1;
    int ch = (hd + want_chunks - 1) / want_chunks;
    if (ch < 2 * R) ch = 2 * R;
    auto cut_rows = [&](int len, int& count, int& height) {
        count = len > 0 ? (len + ch - 1) / ch : 0;
        height = count ? (len + count - 1) / count : 1;
        if (count) count = (len + height - 1) / height;
    };
    cut_rows(sy, g.nya, g.cha);
    cut_rows(hd - sy, g.nyb, g.chb);
    const int ny = g.nya + g.nyb;
    const long long total = (long long)n * nx * ny;
    if (total < min_waves || total > 0x7fffffffLL) return false;
    const int grid = (int)((total + 3) / 4);
    hipExtLaunchKernelGGL((blur_reduce_kernel<R, DCPL>), dim3((unsigned)grid), dim3(256), 0, s, t_rd_start, t_rd_stop, 0, in, dst, w, h, wd, hd,
                          (int)total, g, d_taps);
    return true;
}

template <int R>
static bool launch_reduce_r(hipStream_t s, const float* in, float* dst, int w, int h, int wd, int hd, int n, const float* d_taps, int sx,
                            int sy, int min_waves) {
    // 4 destination columns per lane while that fills the lanes of a strip; 2 for narrow levels (a parity's run of fewer than
    // ~200 columns would leave a 4-wide strip's wave half empty) and for column counts that are not multiples of 4
    const int run = sx > 0 && wd - sx > 0 ? (sx < wd - sx ? sx : wd - sx) : wd;
    const bool wide = run >= 192 || (run % 256) > 128;
    if (wide && launch_reduce_rd<R, 4>(s, in, dst, w, h, wd, hd, n, d_taps, sx, sy, min_waves)) return true;
    return launch_reduce_rd<R, 2>(s, in, dst, w, h, wd, hd, n, d_taps, sx, sy, min_waves);
}

#define SIFT_RD_CASE(R) \
    case R:             \
        return launch_reduce_r<R>(s, in, dst, w, h, wd, hd, n, d_taps, sx, sy, min_waves);

// Blur + decimation, kept pixels only.  sx / sy: first destination column / row whose source index is 2 i + 1 (= wd / hd when
// the map has no such entry); the caller has checked that the index maps have that form.  false: not launched (shape or
// radius outside what the kernel takes) - the caller falls back to blur_stream_kernel<..., DEC> or to blur + resampling.
bool launch_blur_reduce_kept(hipStream_t s, const float* in, float* dst, int w, int h, int wd, int hd, int n, const float* d_taps, int radius,
                             int sx, int sy, int min_waves, hipEvent_t ev_start, hipEvent_t ev_stop) {
    struct Scope {
        Scope(hipEvent_t a, hipEvent_t b) { t_rd_start = a; t_rd_stop = b; }
        ~Scope() { t_rd_start = t_rd_stop = nullptr; }
    } scope(ev_start, ev_stop);
    if (sx < 0 || sx > wd || sy < 0 || sy > hd || wd < 2 || hd < 1) return false;
    switch (radius) {
        SIFT_RD_CASE(2) SIFT_RD_CASE(3) SIFT_RD_CASE(4) SIFT_RD_CASE(5) SIFT_RD_CASE(6) SIFT_RD_CASE(7) SIFT_RD_CASE(8)
        SIFT_RD_CASE(9) SIFT_RD_CASE(10) SIFT_RD_CASE(11) SIFT_RD_CASE(12) SIFT_RD_CASE(13) SIFT_RD_CASE(14) SIFT_RD_CASE(15)
        SIFT_RD_CASE(16) SIFT_RD_CASE(17) SIFT_RD_CASE(18) SIFT_RD_CASE(19) SIFT_RD_CASE(20)
    }
    return false;
}

// The runtime builds a translation unit's device code on the first launch of any of its kernels, and two host threads that make
// their first launches at the same time (several contexts, one thread each) were seen to crash inside that step
// (tools/asan_example.sh: SEGV below hipLaunchKernel).  sift_hip_create touches every unit once, under a lock.
__global__ void tu_probe_reduce_kernel() {}
void tu_touch_reduce(hipStream_t s) { hipLaunchKernelGGL(tu_probe_reduce_kernel, dim3(1), dim3(1), 0, s); }

}  // namespace sift_hip
