// The two "cleanup" steps of Sift::calculate (/root/reference/sift.cpp:37-42 and 49-54) on the GPU:
//     std::sort(points, InterestPoint::cmpByFilter); find first filtered; u16_t size; resize(size)
//
// The order in which libstdc++'s UNSTABLE introsort leaves the surviving points feeds the
// order-dependent descriptor stage, so it must be reproduced exactly.  With a two-valued key
// (filtered or not) the algorithm's behaviour has a closed structure that needs no general sort:
//   * __introsort_loop only ever carries ONE mixed range.  A partition around a "filtered" pivot
//     moves the t-th filtered element from the left to position last-1-t while that is to its
//     right; the right part is then all-filtered (dead: holds no survivor).  A partition around a
//     "kept" pivot moves the t-th kept element from the right to first+t; the left part is then
//     all-kept.
//   * an all-kept range evolves independently of the data (every comparison is false): swap
//     first<->middle, reverse [first+1, last), split — a closed-form position map per element.
//   * __final_insertion_sort is a stable sort, i.e. a stable partition by key: survivors keep the
//     position order they have after the loop.
//   * depth_limit = 2*floor(log2 n); if it ever runs out (heapsort fallback) the image is flagged
//     and the host's std::sort path (host_glue.cpp) redoes it.
// One 1024-thread workgroup per image; ranks come from wavefront __ballot + popcount prefix sums.
// tests/test_abi.py checks the host twin of this scheme, tests/test_gpu_parity.py this kernel,
// against std::sort itself.
#include <type_traits>

#include "common.h"

namespace sift_hip {

constexpr int kCT = 1024;          // threads per cleanup workgroup
constexpr int kMaxPure = 128;      // all-kept ranges spawned by one sort (<= rounds <= 2*31)

struct PureRange {
    int f, m, d;
};

struct CleanupShared {
    int wc[2][4][16];
    int flag;        // first failing position (min) of the current round
    int T;
    int f, l, d, p;
    int npure;
    int fallback;
    PureRange pure[kMaxPure];
};

__device__ __forceinline__ int floor_log2(int n) { return 31 - __clz(n); }

// Key storage.  The keys are read far more often than anything else and every access is on the
// critical path of a one-workgroup-per-image kernel, so they live in LDS as a bit mask whenever
// the image's candidate count fits (kBitCap); otherwise in a global byte array.
struct GlobalKeys {
    uint8_t* k;
    __device__ __forceinline__ int get(int i) const { return k[i]; }
    __device__ __forceinline__ void init(int i, int v) const { k[i] = (uint8_t)v; }
    __device__ __forceinline__ void swap(int a, int b) const {
        const uint8_t ka = k[a], kb = k[b];
        k[a] = kb;
        k[b] = ka;
    }
};
struct LdsBitKeys {
    uint32_t* w;
    __device__ __forceinline__ int get(int i) const { return (int)((w[i >> 5] >> (i & 31)) & 1u); }
    __device__ __forceinline__ void swap(int a, int b) const {  // disjoint pairs may share words: atomics
        if (get(a) != get(b)) {
            atomicXor(&w[a >> 5], 1u << (a & 31));
            atomicXor(&w[b >> 5], 1u << (b & 31));
        }
    }
};
constexpr int kBitCap = 1 << 19;   // 64 KiB of LDS for the key bits
constexpr int kSmallRange = 8192;   // filtered-pivot rounds on ranges this short run element-parallel
constexpr int kMoveSlots = 10;     // pending payload moves per thread (LDS, behind the key bits)
constexpr int kDynLds = kBitCap / 8 + 1024 * kMoveSlots * 8;

// optional phase stamps (diagnostic builds of the KAT entry only): 100 MHz wall clock
__device__ unsigned long long* g_stamp = nullptr;
__device__ __forceinline__ void stamp(int slot) {
    if (kDiagMask && g_stamp && threadIdx.x == 0 && blockIdx.x == 0) g_stamp[slot] = wall_clock64();
}

// Ranks of this thread's four elements (one in each of four consecutive 1024-element chunks, so the
// four loads that produced `hit` are coalesced and in flight together) among the hits of the
// current 4096-element tile, plus the tile's total.  Double-buffered per-wave counters sh.wc[par].
__device__ __forceinline__ void tile_rank4(CleanupShared& sh, int par, const bool (&hit)[4], int (&rank)[4],
                                           int& tile_total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long m[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        m[u] = __ballot(hit[u]);
        if (lane == 0) sh.wc[par][u][wv] = __popcll(m[u]);
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = sh.wc[par][u][q];
            before += (q < wv) ? c : 0;
            total += c;
        }
        rank[u] = run + before + __popcll(m[u] & ((1ull << lane) - 1ull));
        run += total;
    }
    tile_total = run;
}

// Closed-form evolution of one element of an all-kept range [f, f+m) with depth budget d.
// Returns the final position, or -1 if the depth limit would be hit with m > 16.
__device__ __forceinline__ int pure_final_pos(int pos, int f, int m, int d) {
    while (m > 16) {
        if (d == 0) return -1;
        --d;
        const int l = f + m;
        const int mid = f + m / 2;
        if (pos == f) pos = mid;           // __move_median_to_first: every compare false -> swap(first, mid)
        else if (pos == mid) pos = f;
        if (pos >= f + 1) pos = f + l - pos;  // __unguarded_partition with an equal pivot: reverse [f+1, l)
        const int cut = f + ((m & 1) ? (m + 1) / 2 : m / 2);
        if (pos < cut) {
            m = cut - f;
        } else {
            f = cut;
            m = l - cut;
        }
    }
    return pos;
}

// Core: K[0..n) keys (0 kept, 1 filtered) and I[0..n) payload are permuted in global memory like
// __introsort_loop would; I2 receives the arrangement after the all-kept ranges' evolution.
// P is scratch for swap sources (n/2 + 1 entries).  Returns via sh.fallback.
template <class Keys>
__device__ void introsort_binary(CleanupShared& sh, int n, const Keys K, uint32_t* __restrict__ I,
                                 uint32_t* __restrict__ I2, uint32_t* __restrict__ P) {
    const int tid = threadIdx.x;
    stamp(1);
    if (tid == 0) {
        sh.f = 0;
        sh.l = n;
        sh.d = n > 0 ? 2 * floor_log2(n) : 0;
        sh.npure = 0;
        sh.fallback = 0;
    }
    __syncthreads();
    while (true) {
        const int f = sh.f, l = sh.l;
        if (l - f <= 16 || sh.fallback) break;
        __syncthreads();
        if (tid == 0) {
            if (sh.d == 0) {
                sh.fallback = 1;
            } else {
                sh.d -= 1;
                // __move_median_to_first(f, f+1, mid, l-1)
                const int a = f + 1, b = f + (l - f) / 2, c = l - 1;
                const int ka = K.get(a), kb = K.get(b), kc = K.get(c);
                auto comp = [](int x, int y) { return x == 0 && y == 1; };
                int s;
                if (comp(ka, kb)) {
                    if (comp(kb, kc)) s = b;
                    else if (comp(ka, kc)) s = c;
                    else s = a;
                } else if (comp(ka, kc)) s = a;
                else if (comp(kb, kc)) s = c;
                else s = b;
                const int ks = K.get(s);
                K.swap(f, s);
                const uint32_t jf = I[f], js = I[s];
                I[f] = js;
                I[s] = jf;
                sh.p = ks;
                sh.flag = 0x7fffffff;
                sh.T = 0x7fffffff;
            }
        }
        __syncthreads();
        if (sh.fallback) break;
        const int p = sh.p;
        const int F = f + 1, L = l;
        int running = 0;  // hits seen in earlier tiles (block-uniform)
        int par = 0;
        if (p == 1) {
            // t-th filtered element from the left goes to L-1-t while it lies left of it
            for (int base = F; base < L; base += 4 * kCT) {
                bool hit[4];
                int pos[4], rk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    pos[u] = base + u * kCT + tid;
                    hit[u] = pos[u] < L && K.get(pos[u]) == 1;
                }
                int tile_total;
                tile_rank4(sh, par, hit, rk, tile_total);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (hit[u]) {
                        const int t = running + rk[u];
                        if (pos[u] < L - 1 - t) {
                            P[t] = (uint32_t)pos[u];
                        } else {  // participants are a prefix of the hits: the smallest failing rank is T
                            atomicMin(&sh.T, t);
                            atomicMin(&sh.flag, pos[u]);
                        }
                    }
                }
                running += tile_total;
                par ^= 1;
                __syncthreads();
                if (sh.T != 0x7fffffff) break;
            }
        } else {
            // t-th kept element from the right goes to F+t while it lies right of it
            for (int base = L - 1; base >= F; base -= 4 * kCT) {
                bool hit[4];
                int pos[4], rk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    pos[u] = base - u * kCT - tid;
                    hit[u] = pos[u] >= F && K.get(pos[u]) == 0;
                }
                int tile_total;
                tile_rank4(sh, par, hit, rk, tile_total);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (hit[u]) {
                        const int t = running + rk[u];
                        if (F + t < pos[u]) P[t] = (uint32_t)pos[u];
                        else atomicMin(&sh.T, t);
                    }
                }
                running += tile_total;
                par ^= 1;
                __syncthreads();
                if (sh.T != 0x7fffffff) break;
            }
        }
        __syncthreads();
        const int T = (sh.T == 0x7fffffff) ? running : sh.T;
        __syncthreads();
        // apply the T disjoint swaps, four per thread in flight
        for (int t0 = tid; t0 < T; t0 += 4 * kCT) {
            int a[4], b[4];
            uint32_t ia[4], ib[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u * kCT;
                a[u] = t < T ? (int)P[t] : -1;
                b[u] = p == 1 ? (L - 1 - t) : (F + t);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (a[u] >= 0) {
                    ia[u] = I[a[u]];
                    ib[u] = I[b[u]];
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (a[u] >= 0) {
                    K.swap(a[u], b[u]);
                    I[a[u]] = ib[u];
                    I[b[u]] = ia[u];
                }
        }
        __syncthreads();
        if (tid == 0) {
            if (p == 1) {
                const int fail = sh.flag;
                const int cut = (fail != 0x7fffffff && fail == L - 1 - T) ? L - 1 - T : L - T;
                sh.l = cut;  // right part [cut, l) is all filtered: dead
            } else {
                const int cut = F + T;
                if (sh.npure < kMaxPure) {
                    sh.pure[sh.npure] = PureRange{f, cut - f, sh.d};
                    sh.npure += 1;
                } else {
                    sh.fallback = 1;
                }
                sh.f = cut;  // left part [f, cut) is all kept
            }
        }
        __syncthreads();
    }
    __syncthreads();
    stamp(2);
    // Only kept elements are ever read back.  Outside the all-kept ranges they stay where the loop
    // left them; inside, each one moves to its closed-form final position.
#pragma unroll 4
    for (int i = tid; i < n; i += kCT)
        if (K.get(i) == 0) I2[i] = I[i];
    __syncthreads();
    stamp(3);
    const int npure = sh.npure;
    for (int r = 0; r < npure; ++r) {
        const PureRange pr = sh.pure[r];
        if (pr.m <= 16) continue;  // already final
        // four elements per thread: their loads are issued together, the closed forms run meanwhile
        for (int j0 = tid; j0 < pr.m; j0 += 4 * kCT) {
            uint32_t v[4];
            int dest[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * kCT;
                v[u] = j < pr.m ? I[pr.f + j] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * kCT;
                dest[u] = j < pr.m ? pure_final_pos(pr.f + j, pr.f, pr.m, pr.d) : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * kCT;
                if (j < pr.m) {
                    if (dest[u] < 0) sh.fallback = 1;
                    else I2[dest[u]] = v[u];
                }
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// The same rounds for LDS bit keys, word-parallel.  With ~94 % of the candidates filtered almost
// every swap of a round exchanges two filtered elements and changes nothing; only the swaps that
// involve a kept element move a payload (and only kept payloads are ever read again).  So a round
//   1. snapshots the range's 64-bit key words in registers (a contiguous run of words per thread)
//      and counts the hits (filtered keys from the left for a filtered pivot, kept keys from the
//      right for a kept pivot); one workgroup scan gives every thread the rank of its first hit;
//   2. decides per hit whether its swap takes part (pos + t < L-1, resp. F + t < pos: monotone, so
//      the participants are the same prefix the sequential loop stops at) and fetches the swap
//      partners' keys as a bit field (they are a contiguous run of positions: L-1-t, resp. F+t);
//   3. applies only the swaps that matter with atomic bit updates and payload moves.
// Partner positions of different threads never overlap, partner and hit zones are disjoint, and the
// hits come from the snapshot, so no other synchronisation is needed inside a round.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxChunk = kBitCap / 64 / kCT;   // key words per thread when the whole array is one range

__device__ __forceinline__ unsigned long long load_word64(const uint32_t* w, int q) {
    return (unsigned long long)w[2 * q] | ((unsigned long long)w[2 * q + 1] << 32);
}
// keys of positions [lo, lo + len), 1 <= len <= 64: bit j <-> position lo + j
__device__ __forceinline__ unsigned long long load_field(const uint32_t* w, int lo, int len) {
    const int q = lo >> 6, sft = lo & 63;
    unsigned long long v = load_word64(w, q) >> sft;
    if (sft + len > 64) v |= load_word64(w, q + 1) << (64 - sft);
    return len == 64 ? v : (v & ((1ull << len) - 1ull));
}
// position of the i-th (0-based) set bit of h, i < popcount(h)
__device__ __forceinline__ int select64(unsigned long long h, int i) {
    int pos = 0;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        const int c = __popcll((h >> pos) & ((1ull << sft) - 1ull));
        if (i >= c) {
            i -= c;
            pos += sft;
        }
    }
    return pos;
}
// exclusive prefix of v over the workgroup in thread order (+ total); two barriers
__device__ __forceinline__ int block_exclusive_scan(CleanupShared& sh, int v, int& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    __syncthreads();
    if (lane == 63) sh.wc[0][0][wv] = inc;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < kCT / 64; ++q) {
        const int c = sh.wc[0][0][q];
        before += q < wv ? c : 0;
        tot += c;
    }
    total = tot;
    return before + inc - v;
}
__device__ __forceinline__ void key_clear(const LdsBitKeys K, int pos) { atomicAnd(&K.w[pos >> 5], ~(1u << (pos & 31))); }
__device__ __forceinline__ void key_set(const LdsBitKeys K, int pos) { atomicOr(&K.w[pos >> 5], 1u << (pos & 31)); }

// One kept-pivot round, element-parallel: the hits are the KEPT keys, every one of them moves a payload, and in the
// ranges where such rounds happen (the left end fills up with kept elements) whole key words are hits - so the
// swaps are listed by rank (P, global scratch) and applied by all threads, instead of word by word.
// Returns T (block-uniform); sh.T must be 0x7fffffff on entry.
__device__ int round_kept_pivot(CleanupShared& sh, const LdsBitKeys K, uint32_t* __restrict__ I,
                                uint32_t* __restrict__ P, int F, int L) {
    const int tid = threadIdx.x;
    int running = 0, par = 0;
    // t-th kept element from the right goes to F+t while it lies right of it
    for (int base = L - 1; base >= F; base -= 4 * kCT) {
        bool hit[4];
        int pos[4], rk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pos[u] = base - u * kCT - tid;
            hit[u] = pos[u] >= F && K.get(pos[u]) == 0;
        }
        int tile_total;
        tile_rank4(sh, par, hit, rk, tile_total);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (hit[u]) {
                const int t = running + rk[u];
                if (F + t < pos[u]) P[t] = (uint32_t)pos[u];
                else atomicMin(&sh.T, t);
            }
        }
        running += tile_total;
        par ^= 1;
        __syncthreads();
        if (sh.T != 0x7fffffff) break;
    }
    __syncthreads();
    const int T = (sh.T == 0x7fffffff) ? running : sh.T;
    __syncthreads();
    // apply the T disjoint swaps, four per thread in flight
    for (int t0 = tid; t0 < T; t0 += 4 * kCT) {
        int a[4], b[4];
        uint32_t ia[4], ib[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * kCT;
            a[u] = t < T ? (int)P[t] : -1;
            b[u] = F + t;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (a[u] >= 0) {
                ia[u] = I[a[u]];
                ib[u] = I[b[u]];
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (a[u] >= 0) {
                K.swap(a[u], b[u]);
                I[a[u]] = ib[u];
                I[b[u]] = ia[u];
            }
    }
    return T;
}

// One filtered-pivot round, element-parallel (short ranges: the word-parallel form leaves most threads idle
// there).  Only swaps whose partner is kept change anything.  Returns T; sh.T / sh.flag as in the other forms.
__device__ int round_filtered_pivot_small(CleanupShared& sh, const LdsBitKeys K, uint32_t* __restrict__ I,
                                          uint32_t* __restrict__ P, int F, int L) {
    const int tid = threadIdx.x;
    int running = 0, par = 0;
    // t-th filtered element from the left goes to L-1-t while it lies left of it
    for (int base = F; base < L; base += 4 * kCT) {
        bool hit[4];
        int pos[4], rk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pos[u] = base + u * kCT + tid;
            hit[u] = pos[u] < L && K.get(pos[u]) == 1;
        }
        int tile_total;
        tile_rank4(sh, par, hit, rk, tile_total);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (hit[u]) {
                const int t = running + rk[u];
                if (pos[u] < L - 1 - t) {
                    P[t] = (uint32_t)pos[u];
                } else {  // participants are a prefix of the hits: the smallest failing rank is T
                    atomicMin(&sh.T, t);
                    atomicMin(&sh.flag, pos[u]);
                }
            }
        }
        running += tile_total;
        par ^= 1;
        __syncthreads();
        if (sh.T != 0x7fffffff) break;
    }
    __syncthreads();
    const int T = (sh.T == 0x7fffffff) ? running : sh.T;
    __syncthreads();
    for (int t0 = tid; t0 < T; t0 += 4 * kCT) {
        int a[4], b[4];
        uint32_t ib[4];
        bool mv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * kCT;
            b[u] = L - 1 - t;
            mv[u] = t < T && K.get(b[u]) == 0;   // a filtered partner: nothing changes
            a[u] = mv[u] ? (int)P[t] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (mv[u]) ib[u] = I[b[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (mv[u]) {
                key_clear(K, a[u]);   // the kept partner now sits at the hit's position
                key_set(K, b[u]);
                I[a[u]] = ib[u];
            }
    }
    return T;
}

__device__ void introsort_bits(CleanupShared& sh, int n, const LdsBitKeys K, uint32_t* __restrict__ I,
                               uint32_t* __restrict__ I2) {
    uint32_t* __restrict__ P = I2;   // swap-source scratch of the kept-pivot rounds (I2 is only filled after the loop)
    const int tid = threadIdx.x;
    stamp(1);
    if (tid == 0) {
        sh.f = 0;
        sh.l = n;
        sh.d = n > 0 ? 2 * floor_log2(n) : 0;
        sh.npure = 0;
        sh.fallback = 0;
    }
    __syncthreads();
    int round = 0;
    while (true) {
        const int f = sh.f, l = sh.l;
        if (l - f <= 16 || sh.fallback) break;
        __syncthreads();
        if (round < 120) stamp(16 + 4 * round);
        if (tid == 0) {
            if (sh.d == 0) {
                sh.fallback = 1;
            } else {
                sh.d -= 1;
                // __move_median_to_first(f, f+1, mid, l-1)
                const int a = f + 1, b = f + (l - f) / 2, c = l - 1;
                const int ka = K.get(a), kb = K.get(b), kc = K.get(c);
                auto comp = [](int x, int y) { return x == 0 && y == 1; };
                int s;
                if (comp(ka, kb)) {
                    if (comp(kb, kc)) s = b;
                    else if (comp(ka, kc)) s = c;
                    else s = a;
                } else if (comp(ka, kc)) s = a;
                else if (comp(kb, kc)) s = c;
                else s = b;
                const int ks = K.get(s), kf = K.get(f);
                K.swap(f, s);
                if (ks == 0 || kf == 0) {   // a filtered element's payload is never read again
                    const uint32_t jf = I[f], js = I[s];
                    I[f] = js;
                    I[s] = jf;
                }
                sh.p = ks;
                sh.flag = 0x7fffffff;
                sh.T = 0x7fffffff;
            }
        }
        __syncthreads();
        if (sh.fallback) break;
        if (round < 120) stamp(16 + 4 * round + 1);
        const int p = sh.p;
        const int F = f + 1, L = l;
        if (p == 0) {
            const int T0 = round_kept_pivot(sh, K, I, P, F, L);
            __syncthreads();
            if (kDiagMask && round < 120 && g_stamp && tid == 0 && blockIdx.x == 0)
                g_stamp[16 + 4 * round + 3] = (wall_clock64() & 0xffffffffull) | ((unsigned long long)(L - F) << 32);
            ++round;
            if (tid == 0) {
                const int cut = F + T0;
                if (sh.npure < kMaxPure) {
                    sh.pure[sh.npure] = PureRange{f, cut - f, sh.d};
                    sh.npure += 1;
                } else {
                    sh.fallback = 1;
                }
                sh.f = cut;  // left part [f, cut) is all kept
            }
            __syncthreads();
            continue;
        }
        if (L - F <= kSmallRange) {
            const int T1 = round_filtered_pivot_small(sh, K, I, P, F, L);
            __syncthreads();
            if (kDiagMask && round < 120 && g_stamp && tid == 0 && blockIdx.x == 0)
                g_stamp[16 + 4 * round + 3] = (wall_clock64() & 0xffffffffull) | ((unsigned long long)(L - F) << 32) | (1ull << 60);
            ++round;
            if (tid == 0) {
                const int fail = sh.flag;
                const int cut = (fail != 0x7fffffff && fail == L - 1 - T1) ? L - 1 - T1 : L - T1;
                sh.l = cut;  // right part [cut, l) is all filtered: dead
            }
            __syncthreads();
            continue;
        }
        // Filtered pivot on a long range, word-parallel.  (Kept pivots and short ranges were handled above.)
        const int q0 = F >> 6, q1 = (L - 1) >> 6;
        const int nwords = q1 - q0 + 1;
        const int chunk = (nwords + kCT - 1) / kCT;   // <= kMaxChunk
        // 1. snapshot + count.  Thread order = hit order (words ascend from the left), which is what the workgroup
        // scan needs: every thread owns `chunk` consecutive words.
        unsigned long long H[kMaxChunk];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < kMaxChunk; ++j) {
            H[j] = 0ull;
            if (j < chunk) {
                const int q = q0 + tid * chunk + j;
                if (q <= q1) {
                    const int base = q << 6;
                    const int lo = F > base ? F - base : 0;              // first position of the range inside the word
                    const int hi = L - base < 64 ? L - base : 64;        // one past the last
                    unsigned long long m = hi == 64 ? ~0ull : ((1ull << hi) - 1ull);
                    m &= ~0ull << lo;
                    H[j] = load_word64(K.w, q) & m;
                    cnt += __popcll(H[j]);
                }
            }
        }
        int total;
        int t0 = block_exclusive_scan(sh, cnt, total);
        // Only the words left of the point where the two scans meet have swaps to apply, i.e. the first half of the
        // threads in the layout above.  The words are therefore re-dealt round-robin for the second phase; their
        // snapshot and the rank of their first hit travel through the scratch array (3 words each, hit order).
        uint32_t* __restrict__ W = P;
#pragma unroll
        for (int j = 0; j < kMaxChunk; ++j) {
            if (j < chunk) {
                const int wi = tid * chunk + j;
                if (wi < nwords) {
                    W[3 * wi + 0] = (uint32_t)H[j];
                    W[3 * wi + 1] = (uint32_t)(H[j] >> 32);
                    W[3 * wi + 2] = (uint32_t)t0;
                }
                t0 += __popcll(H[j]);
            }
        }
        __syncthreads();
        if (round < 120) stamp(16 + 4 * round + 2);
        int Tw[kMaxChunk];
#pragma unroll
        for (int j = 0; j < kMaxChunk; ++j) {
            H[j] = 0ull;
            Tw[j] = 0;
            if (j < chunk) {
                const int wi = j * kCT + tid;
                if (wi < nwords) {
                    H[j] = (unsigned long long)W[3 * wi + 0] | ((unsigned long long)W[3 * wi + 1] << 32);
                    Tw[j] = (int)W[3 * wi + 2];
                }
            }
        }
        // pending payload moves of this thread: (destination, source) pairs parked in LDS, so that all their loads
        // are in flight together and all stores follow (the moves of a round touch disjoint positions)
        uint2* myq = reinterpret_cast<uint2*>(K.w + kBitCap / 32) + tid * kMoveSlots;
        int nmv = 0;
        auto flush_moves = [&]() {
            uint32_t v[kMoveSlots];
#pragma unroll
            for (int u = 0; u < kMoveSlots; ++u)
                if (u < nmv) v[u] = I[myq[u].y];
#pragma unroll
            for (int u = 0; u < kMoveSlots; ++u)
                if (u < nmv) I[myq[u].x] = v[u];
            nmv = 0;
        };
        // 2 + 3. participation, partner keys, the swaps that matter.  Lanes find different numbers of moves; the
        // queue is drained at wave-uniform points only (when any lane's is full), so a wave pays one load and one
        // store round trip per ~kMoveSlots moves of its busiest lane instead of one per lane and batch.
#pragma unroll
        for (int j = 0; j < kMaxChunk; ++j) {
            const unsigned long long h = H[j];
            const int t = Tw[j];
            const int base = (q0 + j * kCT + tid) << 6;
            const int k = __popcll(h);
            int kp = 0;                  // participating hits of this word (a prefix in hit order)
            unsigned long long M = 0ull; // hits whose partner is kept (they receive it)
            int bref = 0;                // partner of hit 0
            if (j < chunk && h != 0ull) {
                // hit i (ascending position): partner L-1-(t+i); takes part iff pos + t + i < L - 1
                const int pos_last = base + 63 - __clzll((long long)h);
                if (pos_last + t + k - 1 < L - 1) {
                    kp = k;
                } else {
                    unsigned long long r = h;
                    int fpos = 0;
                    while (r) {
                        fpos = base + __ffsll((long long)r) - 1;
                        if (fpos + t + kp >= L - 1) break;
                        ++kp;
                        r &= r - 1ull;
                    }
                    if (t + kp < sh.T) {   // (rank, position) of the first failing hit: both minimal together
                        atomicMin(&sh.T, t + kp);
                        atomicMin(&sh.flag, fpos);
                    }
                }
                if (kp > 0) {
                    bref = L - 1 - t;
                    const unsigned long long field = load_field(K.w, bref - kp + 1, kp);   // bit (kp-1-i) <-> partner of hit i
                    const unsigned long long rev = __brevll(field) >> (64 - kp);
                    M = ~rev & (kp == 64 ? ~0ull : ((1ull << kp) - 1ull));
                }
            }
            while (__any(M != 0ull)) {
                if (M) {
                    const int i = __ffsll((long long)M) - 1;
                    M &= M - 1ull;
                    const int a = base + select64(h, i), b = bref - i;
                    key_clear(K, a);   // the kept element will sit at the hit's position
                    key_set(K, b);
                    myq[nmv++] = make_uint2((unsigned)a, (unsigned)b);   // I[a] = I[b]
                }
                if (__any(nmv == kMoveSlots)) flush_moves();
            }
        }
        flush_moves();
        __syncthreads();
        if (kDiagMask && round < 120 && g_stamp && tid == 0 && blockIdx.x == 0)
            g_stamp[16 + 4 * round + 3] = (wall_clock64() & 0xffffffffull) | ((unsigned long long)(L - F) << 32) | ((unsigned long long)p << 60);
        ++round;
        const int T = (sh.T == 0x7fffffff) ? total : sh.T;
        __syncthreads();
        if (tid == 0) {
            const int fail = sh.flag;
            const int cut = (fail != 0x7fffffff && fail == L - 1 - T) ? L - 1 - T : L - T;
            sh.l = cut;  // right part [cut, l) is all filtered: dead
        }
        __syncthreads();
    }
    __syncthreads();
    if (kDiagMask && g_stamp && tid == 0 && blockIdx.x == 0) g_stamp[7] = (unsigned long long)round;
    stamp(2);
    // Only kept elements are ever read back.  Outside the all-kept ranges they stay where the loop
    // left them; inside, each one moves to its closed-form final position.
    {
        const int nq = (n + 63) >> 6;
        for (int q = tid; q < nq; q += kCT) {
            const int base = q << 6;
            unsigned long long kept = ~load_word64(K.w, q);
            if (n - base < 64) kept &= (1ull << (n - base)) - 1ull;
            while (kept) {
                const int pos = base + __ffsll((long long)kept) - 1;
                kept &= kept - 1ull;
                I2[pos] = I[pos];
            }
        }
    }
    __syncthreads();
    stamp(3);
    const int npure = sh.npure;
    for (int r = 0; r < npure; ++r) {
        const PureRange pr = sh.pure[r];
        if (pr.m <= 16) continue;  // already final
        // four elements per thread: their loads are issued together, the closed forms run meanwhile
        for (int j0 = tid; j0 < pr.m; j0 += 4 * kCT) {
            uint32_t v[4];
            int dest[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * kCT;
                v[u] = j < pr.m ? I[pr.f + j] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * kCT;
                dest[u] = j < pr.m ? pure_final_pos(pr.f + j, pr.f, pr.m, pr.d) : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * kCT;
                if (j < pr.m) {
                    if (dest[u] < 0) sh.fallback = 1;
                    else I2[dest[u]] = v[u];
                }
            }
        }
    }
    __syncthreads();
}

// Stable partition: kept elements in position order -> out[0..size), size = count mod 65536.
// `emit(rank, payload)` is called for rank < size.
template <class Keys, class Emit>
__device__ int compact_kept(CleanupShared& sh, int n, const Keys K, const uint32_t* __restrict__ I2, Emit emit,
                            int* total_out = nullptr) {
    const int tid = threadIdx.x;
    // total first (needed for the u16 truncation)
    int cnt = 0;
#pragma unroll 4
    for (int i = tid; i < n; i += kCT) cnt += K.get(i) == 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (tid == 0) sh.T = 0;
    __syncthreads();
    if ((tid & 63) == 0) atomicAdd(&sh.T, cnt);
    __syncthreads();
    const int total = sh.T;
    const int size = total & 0xffff;  // u16_t size (sift.cpp:41,53)
    if (total_out) *total_out = total;
    __syncthreads();
    int running = 0, par = 0;
    for (int base = 0; base < n && running < size; base += 4 * kCT) {
        bool hit[4];
        int pos[4], rk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pos[u] = base + u * kCT + tid;
            hit[u] = pos[u] < n && K.get(pos[u]) == 0;
        }
        int tile_total;
        tile_rank4(sh, par, hit, rk, tile_total);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (hit[u] && running + rk[u] < size) emit(running + rk[u], I2[pos[u]]);
        running += tile_total;
        par ^= 1;
    }
    __syncthreads();
    return size;
}

extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn_bits[];   // kBitCap / 32 words when the launch provides them

// flags -> key bits (one __ballot per 64 elements, no atomics) and identity payload
__device__ void init_bits_from_flags(const LdsBitKeys K, const uint8_t* __restrict__ fl, int n,
                                     uint32_t* __restrict__ I) {
    const int lane = threadIdx.x & 63;
    constexpr int UN = 8;   // flag loads in flight per thread
    for (int base0 = 0; base0 < n; base0 += UN * kCT) {
        uint8_t v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = base0 + u * kCT + (int)threadIdx.x;
            v[u] = i < n ? fl[i] : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int base = base0 + u * kCT;
            const int i = base + (int)threadIdx.x;
            const unsigned long long m = __ballot(i < n && v[u] != 0);
            if (lane == 0 && i < n) {
                const int wd = (base + ((int)threadIdx.x & ~63)) >> 5;
                K.w[wd] = (uint32_t)m;
                K.w[wd + 1] = (uint32_t)(m >> 32);
            }
            if (i < n) I[i] = (uint32_t)i;
        }
    }
    __syncthreads();
}

template <class Keys>
__device__ void cleanup1_body(CleanupShared& sh, int n, const Keys K, const uint8_t* __restrict__ fl,
                              uint32_t* __restrict__ I, uint32_t* __restrict__ I2, uint32_t* __restrict__ P,
                              uint32_t* __restrict__ out, OrientIn* __restrict__ ord,
                              uint32_t* __restrict__ lrank, const Candidate* __restrict__ cd, int img,
                              int* __restrict__ list_cnt, int* __restrict__ late_cnt, int* __restrict__ fallback) {
    if constexpr (std::is_same<Keys, LdsBitKeys>::value) introsort_bits(sh, n, K, I, I2);
    else introsort_binary(sh, n, K, I, I2, P);
    auto make_in = [&](uint32_t cand, uint32_t kp) {
        const Candidate c = cd[cand];
        OrientIn o;
        o.x = c.x; o.y = c.y; o.octave = c.octave; o.index = c.index; o.kp = kp; o.pad = 0;
        return o;
    };
    int total = 0;
    // I is free from here on: it records each survivor's list position by candidate index
    const int size = compact_kept(sh, n, K, I2, [&](int r, uint32_t id) { out[r] = id; I[id] = (uint32_t)r; }, &total);
    __syncthreads();
    // The (order-independent) orientation stage has run, or is running, on the side stream over ALL
    // kept candidates in ascending candidate index (orient_emit_kernel); its results are addressed
    // by that spatial rank.  lrank[r] = spatial rank of the survivor at list position r.
    if (total == size) {
        int running = 0, par = 0;
        for (int base = 0; base < n; base += 4 * kCT) {
            bool hit[4];
            int pos[4], rk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pos[u] = base + u * kCT + (int)threadIdx.x;
                hit[u] = pos[u] < n && fl[pos[u]] == 0;
            }
            int tile_total;
            tile_rank4(sh, par, hit, rk, tile_total);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (hit[u]) lrank[I[pos[u]]] = (uint32_t)(running + rk[u]);
            running += tile_total;
            par ^= 1;
        }
    } else {  // u16 truncation dropped survivors: the orientation stage runs late, in list order
        for (int r = threadIdx.x; r < size; r += kCT) {
            ord[r] = make_in(out[r], (uint32_t)r);
            lrank[r] = (uint32_t)r;
        }
    }
    if (threadIdx.x == 0) {
        list_cnt[img] = size;
        late_cnt[img] = total == size ? 0 : size;
        fallback[img] = sh.fallback;
    }
}

// LDS-key variant of the first cleanup.  The payload that travels through the sort is the kept
// candidate's SPATIAL RANK (its rank among the kept candidates in candidate order), which is what
// the orientation results are addressed by; rank -> candidate index goes through a table.  Only the
// ~6 % kept candidates get a payload at all, and every pass below walks 64-bit key words.
__device__ void cleanup1_bits_body(CleanupShared& sh, int n, const LdsBitKeys K, const uint8_t* __restrict__ fl,
                                   uint32_t* __restrict__ I, uint32_t* __restrict__ I2, uint32_t* __restrict__ crank,
                                   uint32_t* __restrict__ out, OrientIn* __restrict__ ord,
                                   uint32_t* __restrict__ lrank, const Candidate* __restrict__ cd, int img,
                                   int* __restrict__ list_cnt, int* __restrict__ late_cnt, int* __restrict__ fallback) {
    const int tid = threadIdx.x, lane = tid & 63;
    // flags (bytes 0 / 1) -> key bits.  16 flags per 16-byte load, four loads in flight per thread; the 16
    // key bits of a load are one 16-bit store into the LDS word array.
    {
        unsigned short* kw16 = reinterpret_cast<unsigned short*>(K.w);
        const int n16 = n >> 4;                       // whole 16-flag groups
        const bool vec = (reinterpret_cast<uintptr_t>(fl) & 15u) == 0;
        auto pack4 = [](uint32_t x) {                 // "byte != 0" of each of the 4 bytes -> 4 bits
            const uint32_t m = ((x | ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u;
            return (m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xFu;
        };
        int done = 0;
        if (vec) {
            constexpr int UN = 4;
            for (int g0 = 0; g0 < n16; g0 += UN * kCT) {
                uint4 v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int g = g0 + u * kCT + tid;
                    v[u] = g < n16 ? reinterpret_cast<const uint4*>(fl)[g] : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int g = g0 + u * kCT + tid;
                    if (g < n16)
                        kw16[g] = (unsigned short)(pack4(v[u].x) | (pack4(v[u].y) << 4) | (pack4(v[u].z) << 8) | (pack4(v[u].w) << 12));
                }
            }
            done = n16 << 4;
        }
        // tail (and everything when the flags are not 16-byte aligned): one flag per thread, ballots.  `done` is
        // a multiple of 16; the ballot path wants 64-aligned word pairs, so it restarts at the enclosing multiple
        // of 64 and rewrites those (identical) bits.
        __syncthreads();
        for (int base = done & ~63; base < n; base += kCT) {
            const int i = base + tid;
            const bool f = i < n && fl[i] != 0;
            const unsigned long long m = __ballot(f);
            if (lane == 0 && i < n) {
                const int wd = (base + (tid & ~63)) >> 5;
                K.w[wd] = (uint32_t)m;
                K.w[wd + 1] = (uint32_t)(m >> 32);
            }
        }
        __syncthreads();
    }
    stamp(8);
    const int nq = (n + 63) >> 6;
    const int chunk = (nq + kCT - 1) / kCT;   // <= kMaxChunk
    auto kept_word = [&](int q) {
        unsigned long long kept = ~load_word64(K.w, q);
        const int base = q << 6;
        if (n - base < 64) kept &= (1ull << (n - base)) - 1ull;
        return kept;
    };
    // spatial ranks: payload of every kept candidate + the rank -> candidate table
    int total;
    {
        int cnt = 0;
        for (int j = 0; j < chunk; ++j) {
            const int q = tid * chunk + j;
            if (q < nq) cnt += __popcll(kept_word(q));
        }
        int r = block_exclusive_scan(sh, cnt, total);
        for (int j = 0; j < chunk; ++j) {
            const int q = tid * chunk + j;
            if (q >= nq) break;
            unsigned long long kept = kept_word(q);
            while (kept) {
                const int pos = (q << 6) + __ffsll((long long)kept) - 1;
                kept &= kept - 1ull;
                I[pos] = (uint32_t)r;
                crank[r] = (uint32_t)pos;
                ++r;
            }
        }
        __syncthreads();
    }
    const int size = total & 0xffff;   // u16_t size (sift.cpp:41)
    stamp(9);
    introsort_bits(sh, n, K, I, I2);
    stamp(4);
    // stable partition: kept elements in final position order -> list positions.  The kept elements now sit in
    // the first few hundred words, so the words are dealt round-robin (one per thread and pass) and the
    // positions are first listed by rank (in I, which is dead by now); the dependent gathers
    // position -> spatial rank -> candidate then run evenly over all threads.
    {
        int run = 0;   // kept elements in earlier passes
        for (int q0 = 0; q0 < nq && run < size; q0 += kCT) {
            const int q = q0 + tid;
            unsigned long long kept = q < nq ? kept_word(q) : 0ull;
            int pass_total;
            int r = run + block_exclusive_scan(sh, __popcll(kept), pass_total);
            while (kept && r < size) {
                I[r] = (uint32_t)((q << 6) + __ffsll((long long)kept) - 1);
                kept &= kept - 1ull;
                ++r;
            }
            run += pass_total;
        }
        __syncthreads();
        constexpr int UN = 4;
        for (int r0 = 0; r0 < size; r0 += UN * kCT) {
            uint32_t pos[UN], sr[UN], cand[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int r = r0 + u * kCT + tid;
                pos[u] = r < size ? I[r] : 0u;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) sr[u] = r0 + u * kCT + tid < size ? I2[pos[u]] : 0u;
#pragma unroll
            for (int u = 0; u < UN; ++u) cand[u] = crank[sr[u]];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int r = r0 + u * kCT + tid;
                if (r < size) {
                    out[r] = cand[u];
                    lrank[r] = sr[u];   // the orientation stage's results are addressed by spatial rank
                }
            }
        }
        __syncthreads();
    }
    if (total != size) {  // u16 truncation dropped survivors: the orientation stage runs late, in list order
        for (int r = tid; r < size; r += kCT) {
            const Candidate c = cd[out[r]];
            OrientIn o;
            o.x = c.x; o.y = c.y; o.octave = c.octave; o.index = c.index; o.kp = (uint32_t)r; o.pad = 0;
            ord[r] = o;
            lrank[r] = (uint32_t)r;
        }
    }
    if (tid == 0) {
        list_cnt[img] = size;
        late_cnt[img] = total == size ? 0 : size;
        fallback[img] = sh.fallback;
    }
}

// ---- cleanup 1: flags of the extrema candidates -> ordered survivor list ---------------------
__global__ __launch_bounds__(kCT) void cleanup1_kernel(const uint8_t* __restrict__ flags,
                                                       const int* __restrict__ totals, long long cand_cap,
                                                       uint8_t* __restrict__ wk, uint32_t* __restrict__ wi,
                                                       uint32_t* __restrict__ wi2, uint32_t* __restrict__ wp,
                                                       uint32_t* __restrict__ list,
                                                       OrientIn* __restrict__ oin, uint32_t* __restrict__ lranks,
                                                       const Candidate* __restrict__ cands, int list_cap,
                                                       int* __restrict__ list_cnt, int* __restrict__ late_cnt,
                                                       int* __restrict__ fallback) {
    __shared__ CleanupShared sh;
    const int img = blockIdx.x;
    const int n = totals[img];
    const size_t off = (size_t)img * (size_t)cand_cap;
    uint32_t* I = wi + off;
    uint32_t* I2 = wi2 + off;
    uint32_t* P = wp + off;
    const uint8_t* fl = flags + off;
    uint32_t* out = list + (size_t)img * (size_t)list_cap;
    OrientIn* ord = oin + (size_t)img * (size_t)list_cap;
    uint32_t* lrank = lranks + (size_t)img * (size_t)list_cap;
    const Candidate* cd = cands + off;
    if (n <= kBitCap) {
        const LdsBitKeys K{s_dyn_bits};
        cleanup1_bits_body(sh, n, K, fl, I, I2, P, out, ord, lrank, cd, img, list_cnt, late_cnt, fallback);
    } else {
        const GlobalKeys K{wk + off};
#pragma unroll 4
        for (int i = threadIdx.x; i < n; i += kCT) {
            K.k[i] = fl[i] ? 1 : 0;
            I[i] = (uint32_t)i;
        }
        __syncthreads();
        cleanup1_body(sh, n, K, fl, I, I2, P, out, ord, lrank, cd, img, list_cnt, late_cnt, fallback);
    }
}

__global__ void build_orient_in_kernel(const Candidate* __restrict__ cands, long long cand_cap,
                                       const uint32_t* __restrict__ list, const int* __restrict__ list_cnt,
                                       int list_cap, OrientIn* __restrict__ oin) {
    const int img = blockIdx.y;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= list_cnt[img]) return;
    const Candidate c = cands[(size_t)img * (size_t)cand_cap + list[(size_t)img * (size_t)list_cap + r]];
    OrientIn o;
    o.x = c.x; o.y = c.y; o.octave = c.octave; o.index = c.index; o.kp = (uint32_t)r; o.pad = 0;
    oin[(size_t)img * (size_t)list_cap + r] = o;
}

// ---- orientation stage input, straight from the edge-filter flags ------------------------------
// _orientationAssignment (sift.cpp:163-203) treats every point on its own, so it does not have to wait
// for the first cleanup's ordering: the kept candidates are compacted in ascending candidate index
// (scan order: neighbours share most of their 16x16 window) and processed on the side stream while
// cleanup 1 emulates the sort.  Images whose kept count does not fit the u16_t size of sift.cpp:41
// are left to the late launch that follows cleanup 1 (late_cnt).
constexpr int kOprepChunk = 1024;   // candidates per chunk
constexpr int kOprepGridX = 512;    // chunks are strided over this many workgroups per image

__global__ __launch_bounds__(256) void orient_count_kernel(const uint8_t* __restrict__ flags,
                                                           const int* __restrict__ totals, long long cand_cap,
                                                           int chunks_cap, int* __restrict__ chunk_cnt) {
    __shared__ int s_w[4];
    const int img = blockIdx.y;
    const int n = totals[img];
    const uint8_t* fl = flags + (size_t)img * (size_t)cand_cap;
    for (int chunk = blockIdx.x; chunk * kOprepChunk < n; chunk += gridDim.x) {
        bool hit[4];
        int c = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = chunk * kOprepChunk + u * 256 + (int)threadIdx.x;
            hit[u] = i < n && fl[i] == 0;
            c += __popcll(__ballot(hit[u]));
        }
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) chunk_cnt[(size_t)img * (size_t)chunks_cap + chunk] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void orient_emit_kernel(const uint8_t* __restrict__ flags,
                                                          const int* __restrict__ totals, long long cand_cap,
                                                          int chunks_cap, const int* __restrict__ chunk_cnt,
                                                          const Candidate* __restrict__ cands, int list_cap,
                                                          OrientIn* __restrict__ oin, int* __restrict__ early_cnt) {
    __shared__ int s_red[2][4];
    __shared__ int s_wc[4][4];
    const int img = blockIdx.y;
    const int n = totals[img];
    const int nchunks = (n + kOprepChunk - 1) / kOprepChunk;
    const uint8_t* fl = flags + (size_t)img * (size_t)cand_cap;
    const Candidate* cd = cands + (size_t)img * (size_t)cand_cap;
    const int* cc = chunk_cnt + (size_t)img * (size_t)chunks_cap;
    OrientIn* ord = oin + (size_t)img * (size_t)list_cap;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (nchunks == 0 && blockIdx.x == 0 && threadIdx.x == 0) early_cnt[img] = 0;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        // kept candidates before this chunk, and in the whole image
        int before = 0, all = 0;
        for (int q = threadIdx.x; q < nchunks; q += 256) {
            const int v = cc[q];
            all += v;
            before += q < chunk ? v : 0;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            before += __shfl_xor(before, d);
            all += __shfl_xor(all, d);
        }
        if (lane == 0) { s_red[0][wv] = before; s_red[1][wv] = all; }
        __syncthreads();
        before = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        all = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
        const bool early = all <= 0xFFFF;   // u16_t size = kept count: nothing is truncated
        if (chunk == 0 && threadIdx.x == 0) early_cnt[img] = early ? all : 0;
        if (early) {
            bool hit[4];
            unsigned long long m[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = chunk * kOprepChunk + u * 256 + (int)threadIdx.x;
                hit[u] = i < n && fl[i] == 0;
                m[u] = __ballot(hit[u]);
                if (lane == 0) s_wc[u][wv] = __popcll(m[u]);
            }
            __syncthreads();
            int run = before;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int bw = 0, tot = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int v = s_wc[u][q];
                    bw += q < wv ? v : 0;
                    tot += v;
                }
                if (hit[u]) {
                    const int r = run + bw + __popcll(m[u] & ((1ull << lane) - 1ull));
                    const int i = chunk * kOprepChunk + u * 256 + (int)threadIdx.x;
                    const Candidate c = cd[i];
                    OrientIn o;
                    o.x = c.x; o.y = c.y; o.octave = c.octave; o.index = c.index; o.kp = (uint32_t)r; o.pad = 0;
                    ord[r] = o;
                }
                run += tot;
            }
        }
        __syncthreads();
    }
}

void launch_orient_prepare(hipStream_t s, int n_images, const uint8_t* d_flags, const int* d_totals, long long cand_cap,
                           int* d_chunk_cnt, const Candidate* d_cands, int list_cap, OrientIn* d_oin, int* d_early_cnt) {
    const int chunks_cap = (int)((cand_cap + kOprepChunk - 1) / kOprepChunk);
    const dim3 grid((unsigned)(chunks_cap < kOprepGridX ? chunks_cap : kOprepGridX), (unsigned)n_images);
    hipLaunchKernelGGL(orient_count_kernel, grid, dim3(256), 0, s, d_flags, d_totals, cand_cap, chunks_cap, d_chunk_cnt);
    hipLaunchKernelGGL(orient_emit_kernel, grid, dim3(256), 0, s, d_flags, d_totals, cand_cap, chunks_cap,
                       (const int*)d_chunk_cnt, d_cands, list_cap, d_oin, d_early_cnt);
}

size_t orient_prepare_chunks(long long cand_cap) { return (size_t)((cand_cap + kOprepChunk - 1) / kOprepChunk); }

// ---- cleanup 2: after orientation assignment -> FinalKp list ---------------------------------------
// status[img*4 + {0: count, 1: fallback (depth limit or a point with several orientation peaks),
//                 2: index of the first point whose dead blur throws (or INT_MAX), 3: its code}]
__global__ __launch_bounds__(kCT) void cleanup2_kernel(const Candidate* __restrict__ cands, long long cand_cap,
                                                       const uint32_t* __restrict__ list,
                                                       const int* __restrict__ list_cnt, int list_cap,
                                                       const OrientOut* __restrict__ orient,
                                                       const uint32_t* __restrict__ lranks,
                                                       uint8_t* __restrict__ wk, uint32_t* __restrict__ wi,
                                                       uint32_t* __restrict__ wi2, uint32_t* __restrict__ wp,
                                                       FinalKp* __restrict__ finals, int* __restrict__ final_cnt,
                                                       int* __restrict__ status, FinalKp* __restrict__ recs) {
    __shared__ CleanupShared sh;
    __shared__ int s_multi, s_throw;
    const int img = blockIdx.x;
    const int n = list_cnt[img];
    const size_t off = (size_t)img * (size_t)list_cap;
    // keys as bytes in global memory: for these short, mostly-kept arrays the per-element rounds run ~3x
    // faster on byte keys than on an LDS bit mask (every swap of the mask is two atomics)
    const GlobalKeys K{wk + off};
    uint32_t* I = wi + off;
    uint32_t* I2 = wi2 + off;
    uint32_t* P = wp + off;
    const OrientOut* oo = orient + off;
    const uint32_t* l1 = list + off;
    const Candidate* cd = cands + (size_t)img * (size_t)cand_cap;
    FinalKp* rec = recs + off;   // the record of list position i, gathered once, before the sort
    const uint32_t* lr = lranks + off;   // list position -> index of its orientation result
    if (threadIdx.x == 0) {
        s_multi = 0;
        s_throw = 0x7fffffff;
    }
    __syncthreads();
    stamp(100);
    {
        // two dependent gathers per element (list position -> orientation slot -> result): four elements per
        // thread in flight, and no store inside the loop (on gfx9 a load's wait also waits for older stores)
        constexpr int UN = 4;
        for (int base0 = 0; base0 < n; base0 += UN * kCT) {
            uint32_t slot[UN], cnd[UN];
            OrientOut o[UN];
            Candidate kc[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = base0 + u * kCT + (int)threadIdx.x;
                slot[u] = i < n ? lr[i] : 0u;
                cnd[u] = i < n ? l1[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = base0 + u * kCT + (int)threadIdx.x;
                o[u].filtered = 0; o[u].npeaks = 0; o[u].throws = 0; o[u].orientation = 0.0f;
                kc[u].x = kc[u].y = kc[u].octave = kc[u].index = 0;
                if (i < n) {
                    o[u] = oo[slot[u]];
                    kc[u] = cd[cnd[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int base = base0 + u * kCT;
                const int i = base + (int)threadIdx.x;
                if (i < n) {
                    FinalKp f;
                    f.cand = cnd[u];
                    f.orientation = o[u].orientation;
                    f.x = kc[u].x; f.y = kc[u].y; f.octave = kc[u].octave; f.index = kc[u].index;
                    rec[i] = f;
                    K.k[i] = o[u].filtered != 0 ? 1 : 0;
                    if (!o[u].filtered && o[u].npeaks > 1) s_multi = 1;
                    if (!o[u].filtered && o[u].throws) atomicMin(&s_throw, i);
                }
            }
        }
        for (int i = threadIdx.x; i < n; i += kCT) I[i] = (uint32_t)i;
    }
    __syncthreads();
    // kept keys dominate here (few points fail the border test): every hit moves a payload, which the
    // tile-based rounds spread over all threads
    stamp(101);
    introsort_binary(sh, n, K, I, I2, P);
    stamp(102);
    FinalKp* out = finals + off;
    const int size = compact_kept(sh, n, K, I2, [&](int r, uint32_t id) { out[r] = rec[id]; });
    stamp(103);
    stamp(104);
    if (threadIdx.x == 0) {
        const bool thr = s_throw != 0x7fffffff;
        final_cnt[img] = thr ? 0 : size;
        status[img * 4 + 0] = thr ? 0 : size;
        status[img * 4 + 1] = (sh.fallback || s_multi) ? 1 : 0;
        status[img * 4 + 2] = s_throw;
        status[img * 4 + 3] = thr ? oo[lr[s_throw]].throws : 0;
    }
}

// KAT entry: flags -> survivor order (whole kept prefix, no u16 truncation applied by the caller)
__global__ __launch_bounds__(kCT) void cleanup_kat_kernel(const uint8_t* __restrict__ flags, int n,
                                                          uint8_t* __restrict__ Kg, uint32_t* __restrict__ I,
                                                          uint32_t* __restrict__ I2, uint32_t* __restrict__ P,
                                                          uint32_t* __restrict__ out, int* __restrict__ info,
                                                          int force_global, OrientIn* __restrict__ ord,
                                                          uint32_t* __restrict__ lrank,
                                                          const Candidate* __restrict__ cd) {
    __shared__ CleanupShared sh;
    __shared__ int s_cnt[3];
    int size;
    stamp(0);
    if (n <= kBitCap && !force_global) {
        // the production body of the first cleanup (its list is the survivor order this entry returns)
        const LdsBitKeys K{s_dyn_bits};
        cleanup1_bits_body(sh, n, K, flags, I, I2, P, out, ord, lrank, cd, 0, &s_cnt[0], &s_cnt[1], &s_cnt[2]);
        __syncthreads();
        size = s_cnt[0];
        stamp(5);
        if (kDiagMask && threadIdx.x == 0 && g_stamp) g_stamp[6] = (unsigned long long)sh.npure;
    } else {
        const GlobalKeys K{Kg};
        for (int i = threadIdx.x; i < n; i += kCT) {
            K.k[i] = flags[i] ? 1 : 0;
            I[i] = (uint32_t)i;
        }
        __syncthreads();
        introsort_binary(sh, n, K, I, I2, P);
        size = compact_kept(sh, n, K, I2, [&](int r, uint32_t id) { out[r] = id; });
    }
    if (threadIdx.x == 0) {
        info[0] = size;
        info[1] = sh.fallback;
    }
}

void launch_cleanup1(hipStream_t s, int n_images, const uint8_t* d_flags, const int* d_totals, long long cand_cap,
                     uint8_t* wk, uint32_t* wi, uint32_t* wi2, uint32_t* wp, uint32_t* d_list, OrientIn* d_oin,
                     uint32_t* d_lrank, const Candidate* d_cands, int list_cap, int* d_list_cnt, int* d_late_cnt,
                     int* d_fallback) {
    static const bool attr_ok = [] {
        LaunchGuard guard;   // creates the kernel's function object: never beside another thread's launch (launch_guard.h)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(cleanup1_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kDynLds) == hipSuccess;
    }();
    (void)attr_ok;
    hipLaunchKernelGGL(cleanup1_kernel, dim3((unsigned)n_images), dim3(kCT), kDynLds, s, d_flags, d_totals, cand_cap,
                       wk, wi, wi2, wp, d_list, d_oin, d_lrank, d_cands, list_cap, d_list_cnt, d_late_cnt, d_fallback);
}

void launch_build_orient_in(hipStream_t s, const Candidate* d_cands, long long cand_cap, const uint32_t* d_list,
                            const int* d_list_cnt, int list_cap, int n_images, OrientIn* d_oin) {
    hipLaunchKernelGGL(build_orient_in_kernel, dim3((unsigned)((list_cap + 255) / 256), (unsigned)n_images), dim3(256), 0,
                       s, d_cands, cand_cap, d_list, d_list_cnt, list_cap, d_oin);
}

void launch_cleanup2(hipStream_t s, int n_images, const Candidate* d_cands, long long cand_cap,
                     const uint32_t* d_list, const int* d_list_cnt, int list_cap, const OrientOut* d_orient,
                     const uint32_t* d_lrank, uint8_t* wk, uint32_t* wi, uint32_t* wi2, uint32_t* wp, FinalKp* d_final,
                     int* d_final_cnt, int* d_status, FinalKp* d_recs) {
    hipLaunchKernelGGL(cleanup2_kernel, dim3((unsigned)n_images), dim3(kCT), 0, s, d_cands, cand_cap, d_list,
                       d_list_cnt, list_cap, d_orient, d_lrank, wk, wi, wi2, wp, d_final, d_final_cnt, d_status, d_recs);
}

void cleanup_set_stamp_buffer(unsigned long long* d) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), &d, sizeof(d)); }

void launch_cleanup_kat(hipStream_t s, const uint8_t* d_flags, int n, uint8_t* wk, uint32_t* wi, uint32_t* wi2,
                        uint32_t* wp, uint32_t* d_out, int* d_info, int force_global, OrientIn* d_ord, uint32_t* d_lrank,
                        const Candidate* d_cd) {
    static const bool attr_ok = [] {
        LaunchGuard guard;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(cleanup_kat_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kDynLds) == hipSuccess;
    }();
    (void)attr_ok;
    hipLaunchKernelGGL(cleanup_kat_kernel, dim3(1), dim3(kCT), kDynLds, s, d_flags, n, wk, wi, wi2, wp, d_out, d_info,
                       force_global, d_ord, d_lrank, d_cd);
}

// The runtime builds a translation unit's device code on the first launch of any of its kernels, and two host threads that make
// their first launches at the same time (several contexts, one thread each) were seen to crash inside that step
// (tools/asan_example.sh: SEGV below hipLaunchKernel).  sift_hip_create touches every unit once, under a lock.
__global__ void tu_probe_cleanup_kernel() {}
void tu_touch_cleanup(hipStream_t s) { hipLaunchKernelGGL(tu_probe_cleanup_kernel, dim3(1), dim3(1), 0, s); }

}  // namespace sift_hip
