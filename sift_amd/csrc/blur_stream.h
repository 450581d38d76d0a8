// The streaming blur kernel of the pyramid (alg::convolveWithGauss, /root/reference/algorithms.cpp:10-22, with alg::dog,
// :52-64, and the decimation of alg::reduceToNextLevel, :24-36), as a header: sift_amd/csrc/kernels_pyramid.hip launches it,
// tools/probe/blur_probe.hip times it alone on the bench's launch shapes.  Device code only; needs <hip/hip_runtime.h>.
#pragma once

#pragma clang fp contract(off)

namespace sift_hip {

__device__ __forceinline__ int reflect_clamp(int p, int n) {
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * (n - 1) - p : p;
    // lanes that only feed outputs outside the image may still be out of range: keep them legal
    return p < 0 ? 0 : (p >= n ? n - 1 : p);
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
// Streaming blur, rows in PAIRS (round 6; radii 6 .. 14, no DoG output, no decimation: the form every level launch of the
// default plan takes).  Same wave-per-strip walk and the same arithmetic per output pixel as blur_stream_kernel - the row
// pass's ascending-order sum from 0.0f, the column sums that slide through registers - but the packed instructions of the
// ROW pass pair two consecutive ROWS of one column instead of two adjacent columns of one row:
//   * a wave keeps rows t and t+1 INTERLEAVED in its LDS line (slot i of row t at float 2i, of row t+1 at 2i+1), so one
//     16-byte read hands a lane the aligned register pairs (row t, row t+1) of two neighbouring columns.  Every term
//     tap[k] * src[x-R+k] of both rows is then ONE v_pk_mul_f32 on an aligned pair whatever the parity of x-R+k.  Pairing
//     adjacent columns (rounds 1 - 5) needs the pair (v[i], v[i+1]) for every i, and half of those start on an odd register:
//     ~15 of the 92 vector instructions per row at R = 10 were moves that lined them up (profiles/r06_blur_isa.txt);
//   * a lane's two columns are two INDEPENDENT chains of 2R+1 dependent additions where the column pairing had one;
//   * the two row-pass results per column are re-paired by column (two v_pk_mov_b32 per row pair) for the column pass,
//     which is the old one, row t then row t+1;
//   * source rows and output rows are addressed through buffer descriptors - a wave-uniform row offset in an SGPR plus the
//     lane's fixed byte offset - so no per-row 64-bit vector address arithmetic is left, and while the rows a loop body
//     fetches lie inside the image the row offset is a running sum (the reflection arithmetic, ~14 scalar instructions per
//     row, only runs in the run-in and at the bottom of an image).
// Bit-identical to blur_stream_kernel by construction (the same operations in the same order per output); the probe and
// the parity tests check it.
// ---------------------------------------------------------------------------------------------
constexpr int stream2_occ(int r) { return r <= 10 ? 3 : 2; }   // (R 11, 12 would spill at the 168 registers three waves leave)

template <int R, int VAR = 0>
__global__ __launch_bounds__(256, stream2_occ(R)) void blur_stream2_kernel(const float* __restrict__ in, float* __restrict__ out, int w,
                                                                           int h, int strips, int strip_w, int chunks, int chunk_h,
                                                                           int total_units, const float* __restrict__ taps) {
    constexpr int PF = kStreamPF;               // source rows in flight per wave: two row pairs
    static_assert(PF == 4, "two row pairs in flight");
    constexpr int RI = stream_runin(R);
    constexpr int E = RI - 2 * R;
    constexpr int RA = (R + 1) & ~1;            // halo columns each side, a whole number of lane pairs
    constexpr int PAD = RA - R;
    constexpr int NT = 2 * R + 1;
    constexpr int ROWF = 128 + 2 * RA;          // column slots per row
    constexpr int NQ = RA + 1;                  // 16-byte window reads per lane: slots 2 el .. 2 el + 2 RA + 1
    __shared__ __attribute__((aligned(16))) float s_rows[4][2 * ROWF];
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));

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
    const int y0 = min(chunk * chunk_h, h - chunk_h);   // the last chunk is pulled up (it rewrites a few rows with the same values)
    const int nsteps = chunk_h + RI;
    const int p0 = y0 - R - E;                          // source row of stream index 0 (reflected)
    float* ring = s_rows[wave];

    const int el = min(lane, sw / 2 - 1);               // lanes beyond the strip shadow its last lane
    const int mcol = xs + 2 * el;
    const bool has_halo = lane < 2 * RA;
    const int hl = has_halo ? lane : 0;
    const int hcol = reflect_clamp(hl < RA ? xs - RA + hl : xs + sw + (hl - RA), w);
    const int hslot = hl < RA ? hl : RA + sw + (hl - RA);
    const unsigned moff = 4u * (unsigned)mcol, hoff = 4u * (unsigned)hcol;
    const int rowbytes = 4 * w;
    const size_t img_floats = (size_t)w * (size_t)h;
    // raw buffer descriptors over this image of the source and of the destination level (flags: gfx9 32-bit raw buffer)
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)img * img_floats), 0, (int)(img_floats * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)img * img_floats, 0, (int)(img_floats * 4), 0x00020000);

    float tp[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) tp[k] = taps[k];
    f2 A[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) A[j] = (f2)(0.0f);

    u2 pmA[2], pmB[2];       // rows T and T+1 of the pair in prefetch slot U: the lane's two columns
    unsigned phA[2], phB[2]; // ... and its halo column
    f4 W[NQ];                // window of the pair the next row pass takes: (slot 2c: row T, row T+1; slot 2c+1: row T, row T+1)
    int prow = 0;            // PLAIN loop: byte offset of the next source row to fetch
    int orow = 0;            // byte offset of the next output row

    // rows T, T+1 of the stream into slot U.  PLAIN: both lie inside the image, offsets run; else reflected / clamped
#define SIFT_S2_FETCH(T, U, PLAIN)                                                                                   \
    {                                                                                                                \
        int sa_, sb_;                                                                                                \
        if (PLAIN) { sa_ = prow; sb_ = prow + rowbytes; prow += 2 * rowbytes; }                                     \
        else { sa_ = reflect_clamp(p0 + (T), h) * rowbytes; sb_ = reflect_clamp(p0 + (T) + 1, h) * rowbytes; }     \
        pmA[U] = __builtin_amdgcn_raw_buffer_load_b64(rs_in, moff, sa_, 0);                                          \
        phA[U] = __builtin_amdgcn_raw_buffer_load_b32(rs_in, hoff, sa_, 0);                                          \
        pmB[U] = __builtin_amdgcn_raw_buffer_load_b64(rs_in, moff, sb_, 0);                                          \
        phB[U] = __builtin_amdgcn_raw_buffer_load_b32(rs_in, hoff, sb_, 0);                                          \
    }
    // LDS stage of the pair in slot U (stream rows T, T+1): interleaved into the line, slot U refilled from HBM, window read
#define SIFT_S2_LDS(T, U, PLAIN)                                                                                     \
    {                                                                                                                \
        const int mi_ = 2 * RA + 4 * el;                                                                             \
        ring[mi_ + 0] = __uint_as_float(pmA[U].x);                                                                   \
        ring[mi_ + 1] = __uint_as_float(pmB[U].x);                                                                   \
        ring[mi_ + 2] = __uint_as_float(pmA[U].y);                                                                   \
        ring[mi_ + 3] = __uint_as_float(pmB[U].y);                                                                   \
        if (has_halo) {                                                                                              \
            ring[2 * hslot + 0] = __uint_as_float(phA[U]);                                                           \
            ring[2 * hslot + 1] = __uint_as_float(phB[U]);                                                           \
        }                                                                                                            \
        __builtin_amdgcn_wave_barrier();                                                                             \
        SIFT_S2_FETCH((T) + PF, U, PLAIN)                                                                            \
        const f4* p4_ = reinterpret_cast<const f4*>(ring) + el;                                                      \
        _Pragma("unroll") for (int c = 0; c < NQ; ++c) W[c] = p4_[c];                                                \
    }
    // column pass of one row: the sliding sums take the row-pass result m (the lane's two columns); A[0] is then complete
#define SIFT_S2_COL(M, STORE)                                                                                        \
    {                                                                                                                \
        f2 An[NT];                                                                                                   \
        _Pragma("unroll") for (int i = 0; i <= R; ++i) {                                                             \
            const f2 pr = tp[i] * (M);                                                                               \
            An[i] = A[i + 1] + pr;                                                                                   \
            if (2 * R - i != i) An[2 * R - i] = (2 * R - i + 1 < NT ? A[2 * R - i + 1] : (f2)(0.0f)) + pr;           \
        }                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) A[j] = An[j];                                                 \
        if (STORE) {                                                                                                 \
            const u2 o_ = {__float_as_uint(A[0].x), __float_as_uint(A[0].y)};                                        \
            __builtin_amdgcn_raw_buffer_store_b64(o_, rs_out, moff, orow, 2 /* nt */);                               \
            orow += rowbytes;                                                                                        \
        }                                                                                                            \
    }
    // one pair of rows S, S+1 (window in W): row pass of both, the next pair's LDS stage (its latency hides under the column
    // passes), re-pairing by column, the two column passes
#define SIFT_S2_PAIR(S, U, STORE, PLAIN)                                                                             \
    {                                                                                                                \
        f2 accA = (f2)(0.0f), accB = (f2)(0.0f);                                                                     \
        {                                                                                                            \
            f2 v[2 * NQ];                                                                                            \
            _Pragma("unroll") for (int c = 0; c < NQ; ++c) {                                                         \
                v[2 * c] = (f2){W[c].x, W[c].y};                                                                     \
                v[2 * c + 1] = (f2){W[c].z, W[c].w};                                                                 \
            }                                                                                                        \
            _Pragma("unroll") for (int k = 0; k < NT; ++k) {                                                         \
                const float tap = tp[NT - 1 - k];                                                                    \
                accA += tap * v[PAD + k];                                                                            \
                accB += tap * v[PAD + k + 1];                                                                        \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_wave_barrier();                                                                             \
        SIFT_S2_LDS((S) + 2, (U) ^ 1, PLAIN)                                                                         \
        const f2 m0_ = {accA.x, accB.x}, m1_ = {accA.y, accB.y};                                                     \
        SIFT_S2_COL(m0_, STORE)                                                                                      \
        SIFT_S2_COL(m1_, STORE)                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }

    SIFT_S2_FETCH(0, 0, false)
    SIFT_S2_FETCH(2, 1, false)
    SIFT_S2_LDS(0, 0, false)
    int s0 = 0;
    // run-in: RI rows that only feed the partial sums
#pragma unroll 1
    for (; s0 < RI; s0 += 4) {
        SIFT_S2_PAIR(s0, 0, false, false)
        SIFT_S2_PAIR(s0 + 2, 1, false, false)
    }
    orow = y0 * rowbytes;
    // steady state, rows fetched inside the image: a body at s0 fetches the stream rows s0 + 6 .. s0 + 9
    if (VAR & 1) {
        prow = (p0 + s0 + 6) * rowbytes;
#pragma unroll 1
        for (; s0 < nsteps && p0 + s0 + 9 < h; s0 += 4) {
            SIFT_S2_PAIR(s0, 0, true, true)
            SIFT_S2_PAIR(s0 + 2, 1, true, true)
        }
    }
    // ... and the rest (the bottom of the image: reflected rows; rows past the stream's end are clamped and never used)
#pragma unroll 1
    for (; s0 < nsteps; s0 += 4) {
        SIFT_S2_PAIR(s0, 0, true, false)
        SIFT_S2_PAIR(s0 + 2, 1, true, false)
    }
#undef SIFT_S2_PAIR
#undef SIFT_S2_COL
#undef SIFT_S2_LDS
#undef SIFT_S2_FETCH
}

}  // namespace sift_hip
