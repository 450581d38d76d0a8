// Wire format of the keypoint gather (only keypoint lists travel between GPUs).
//
// A descriptor (`InterestPoint::descriptors`, 128 floats, sift.cpp:55 / algorithms.cpp:118-151) never has mass in
// bin 7 of a cell (`% 7`, algorithms.cpp:135-150) and only in the bins its 16 samples fall into: on real frames
// about a third of the remaining 112 floats are set.  On the wire a keypoint is therefore
//     34 bytes   its 20-byte record + 112 presence bits (bit j <-> float cell*8+bin, j = cell*7+bin)
//     4 bytes    per float whose bit pattern is not +0.0f, in ascending position
// — lossless; sift_amd/gather.py:unpack_sparse restores the 128 floats bit for bit.
//
// Two passes over the result arrays (the per-keypoint offsets need a scan): counts per block of 64 keypoints, a
// one-workgroup scan of the block sums, then the emit pass.  A wave reads one descriptor as 64 float2: 512
// contiguous bytes.  Both passes run on every rank of a multi-GPU job once per batch, beside the next batch's kernels.
#include "common.h"

namespace sift_hip {

constexpr int kWireBlock = 64;   // keypoints per workgroup: 4 waves x 16
constexpr int kWirePerWave = kWireBlock / 4;

__device__ __forceinline__ void wire_presence(const float2 v, int lane, bool& b0, bool& b1) {
    // positions 2*lane and 2*lane+1; bin = position % 8; bin 7 never carries information
    b0 = __float_as_uint(v.x) != 0u;                                  // 2*lane is even: its bin is never 7
    b1 = (((2 * lane + 1) & 7) != 7) && __float_as_uint(v.y) != 0u;
}

// A wave takes kWirePerWave consecutive descriptors: all their loads are issued before the first is looked at (16 x 512 B
// in flight per wave), the presence bits come from two wavefront ballots per descriptor (even / odd positions).

// *flag (cleared by the launcher) is raised when some descriptor carries anything but +0.0f in bin 7 of a
// cell: normalizeVector divides every bin by the cell's sum (algorithms.cpp:210-223), so a negative or NaN sum (negative
// pixels, inf * 0 after the cumulative `magnitudes += weighting`) turns the never-written bin into -0.0f or NaN, which
// this format cannot carry
__global__ __launch_bounds__(256) void wire_count_kernel(const float* __restrict__ desc, long long total,
                                                        int* __restrict__ block_sums, int* __restrict__ flag) {
    __shared__ int s_sum[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long k0 = (long long)blockIdx.x * kWireBlock + wave * kWirePerWave;
    float2 v[kWirePerWave];
#pragma unroll
    for (int i = 0; i < kWirePerWave; ++i) {
        const long long k = min(k0 + i, total - 1);   // past the end: the last descriptor again (not counted)
        v[i] = reinterpret_cast<const float2*>(desc + k * 128)[lane];
    }
    int sum = 0;
    unsigned long long bin7 = 0ull;
#pragma unroll
    for (int i = 0; i < kWirePerWave; ++i) {
        bool b0, b1;
        wire_presence(v[i], lane, b0, b1);
        const int n = __popcll(__ballot(b0)) + __popcll(__ballot(b1));
        sum += (k0 + i < total) ? n : 0;
        bin7 |= __ballot((lane & 3) == 3 && __float_as_uint(v[i].y) != 0u);
    }
    if (bin7 != 0ull && lane == 0) atomicOr(flag, 1);
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// exclusive scan of the block sums (one workgroup); block_off[nb] = total number of floats
__global__ __launch_bounds__(1024) void wire_scan_kernel(const int* __restrict__ block_sums, int nb,
                                                         long long* __restrict__ block_off) {
    __shared__ long long s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nb + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, nb);
    long long sum = 0;
    for (int i = lo; i < hi; ++i) sum += block_sums[i];
    // inclusive scan inside the wave (shuffles), then over the 16 wave totals
    long long inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    long long before = 0;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    long long run = before + inc - sum;
    for (int i = lo; i < hi; ++i) {
        block_off[i] = run;
        run += block_sums[i];
    }
    if (tid == 1023) block_off[nb] = before + inc;
}

// Presence bit j = cell * 7 + bin of a descriptor whose even / odd positions are set in `even` / `odd` (bit L <-> positions
// 2L / 2L + 1): position p = cell * 8 + bin.
__device__ __forceinline__ bool wire_bit(int j, unsigned long long even, unsigned long long odd) {
    const int p = (j / 7) * 8 + (j % 7);
    return (((p & 1) ? odd : even) >> (p >> 1)) & 1ull;
}

__global__ __launch_bounds__(256) void wire_emit_kernel(const sift_hip_keypoint* __restrict__ kp,
                                                       const float* __restrict__ desc, long long total,
                                                       const long long* __restrict__ block_off,
                                                       uint8_t* __restrict__ records, float* __restrict__ values) {
    __shared__ int s_sum[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const long long k0 = (long long)blockIdx.x * kWireBlock + wave * kWirePerWave;
    float2 v[kWirePerWave];
#pragma unroll
    for (int i = 0; i < kWirePerWave; ++i) {
        const long long k = min(k0 + i, total - 1);
        v[i] = reinterpret_cast<const float2*>(desc + k * 128)[lane];
    }
    // the keypoint records of this wave's descriptors as 16-bit words (a wire record is 34 bytes: only 2-byte aligned):
    // lane (i, s) = descriptor i / 4 + 4 * round, word s < 10  -> four loads cover the 16 records
    const unsigned short* __restrict__ kps = reinterpret_cast<const unsigned short*>(kp);
    unsigned short rw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + (lane >> 4), sidx = lane & 15;
        const long long k = min(k0 + i, total - 1);
        rw[r] = sidx < 10 ? kps[k * 10 + sidx] : (unsigned short)0;
    }
    unsigned long long even[kWirePerWave], odd[kWirePerWave];
    int sum = 0;
#pragma unroll
    for (int i = 0; i < kWirePerWave; ++i) {
        bool b0, b1;
        wire_presence(v[i], lane, b0, b1);
        even[i] = __ballot(b0);
        odd[i] = __ballot(b1);
        sum += (k0 + i < total) ? __popcll(even[i]) + __popcll(odd[i]) : 0;
    }
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    long long base = block_off[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += s_sum[w];
    unsigned short* __restrict__ rec16 = reinterpret_cast<unsigned short*>(records);
    // presence bits of this lane's two positions j = lane and lane + 64 in the 112-bit mask: constant per lane
#pragma unroll
    for (int i = 0; i < kWirePerWave; ++i) {
        if (k0 + i >= total) break;   // wave-uniform
        // values, ascending position: lane's even element, then its odd one
        const int r = __popcll(even[i] & below) + __popcll(odd[i] & below);
        const bool b0 = (even[i] >> lane) & 1ull, b1 = (odd[i] >> lane) & 1ull;
        if (b0) values[base + r] = v[i].x;
        if (b1) values[base + r + (b0 ? 1 : 0)] = v[i].y;
        base += __popcll(even[i]) + __popcll(odd[i]);
        // mask words: bit j of the 112 from two ballots
        const unsigned long long m0 = __ballot(wire_bit(lane, even[i], odd[i]));
        const unsigned long long m1 = __ballot(lane < 48 && wire_bit(lane + 64, even[i], odd[i]));
        // 17 words of the record: 10 of the keypoint (held by lane 16 * (i % 4) + s of load round i / 4), 7 of the mask
        const unsigned short kw = (unsigned short)__shfl((int)rw[i >> 2], 16 * (i & 3) + (lane & 15));
        unsigned short word = kw;
        if (lane >= 10) {
            const int t = lane - 10;   // 0..6: mask bits 16 t .. 16 t + 15
            word = (unsigned short)((t < 4 ? (m0 >> (16 * t)) : (m1 >> (16 * (t - 4)))) & 0xffffull);
        }
        if (lane < 17) rec16[(k0 + i) * 17 + lane] = word;
    }
}

// ---- the receiving side: 34-byte records + set floats -> sift_hip_keypoint records + 128-float descriptors ----------------
// floats per block of 64 records, from the presence bits alone
__global__ __launch_bounds__(256) void wire_mask_count_kernel(const uint8_t* __restrict__ records, long long total,
                                                             int* __restrict__ block_sums) {
    __shared__ int s_sum[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long k = (long long)blockIdx.x * kWireBlock + threadIdx.x / 4;   // four lanes per record
    const unsigned short* __restrict__ r16 = reinterpret_cast<const unsigned short*>(records);
    int n = 0;
    if (k < total) {
        // mask = words 10..16 of the record: lane q of the four takes words 10 + q and 14 + q (q < 3)
        const int q = lane & 3;
        n = __popc((unsigned)r16[k * 17 + 10 + q]) + (q < 3 ? __popc((unsigned)r16[k * 17 + 14 + q]) : 0);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) s_sum[wave] = n;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

__global__ __launch_bounds__(256) void wire_unpack_kernel(const uint8_t* __restrict__ records, const float* __restrict__ values,
                                                         long long total, const long long* __restrict__ block_off,
                                                         sift_hip_keypoint* __restrict__ kp, float* __restrict__ desc) {
    __shared__ int s_sum[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long k0 = (long long)blockIdx.x * kWireBlock + wave * kWirePerWave;
    const unsigned short* __restrict__ r16 = reinterpret_cast<const unsigned short*>(records);
    // the wave's 16 records, 17 words each, as 16-bit loads: lane (i, s) = record i of four per round, word s
    // round r covers records 4r .. 4r+3: lane 16 * (record in round) + s holds word s < 16, every lane of the 16 word 16
    unsigned short w16[4], wlast[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long k = min(k0 + 4 * r + (lane >> 4), total - 1);
        w16[r] = r16[k * 17 + (lane & 15)];
        wlast[r] = r16[k * 17 + 16];
    }
    unsigned long long m0[kWirePerWave], m1[kWirePerWave];
    int sum = 0;
#pragma unroll
    for (int i = 0; i < kWirePerWave; ++i) {
        const int src = 16 * (i & 3);
        unsigned long long a = 0ull, b = 0ull;
#pragma unroll
        for (int t = 0; t < 4; ++t) a |= (unsigned long long)(unsigned short)__shfl((int)w16[i >> 2], src + 10 + t) << (16 * t);
        b = (unsigned long long)(unsigned short)__shfl((int)w16[i >> 2], src + 14) |
            ((unsigned long long)(unsigned short)__shfl((int)w16[i >> 2], src + 15) << 16) |
            ((unsigned long long)(unsigned short)__shfl((int)wlast[i >> 2], src) << 32);
        m0[i] = a; m1[i] = b;
        sum += (k0 + i < total) ? __popcll(a) + __popcll(b) : 0;
    }
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    long long base = block_off[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += s_sum[w];
    unsigned short* __restrict__ kp16 = reinterpret_cast<unsigned short*>(kp);
    // this lane's positions 2 * lane and 2 * lane + 1: cell = position / 8, bin = position % 8; presence bit j = cell * 7 + bin
    const int p0 = 2 * lane, p1 = 2 * lane + 1;
    const int j0 = (p0 >> 3) * 7 + (p0 & 7), j1 = (p1 >> 3) * 7 + (p1 & 7);
    const bool odd7 = (p1 & 7) == 7;   // bin 7 is never on the wire: +0.0f
    auto below = [](unsigned long long a, unsigned long long b, int j) {   // set bits among presence bits 0 .. j-1
        return j < 64 ? __popcll(a & ((1ull << j) - 1ull)) : __popcll(a) + __popcll(b & ((1ull << (j - 64)) - 1ull));
    };
    auto bit = [](unsigned long long a, unsigned long long b, int j) { return (int)(((j < 64 ? a >> j : b >> (j - 64))) & 1ull); };
#pragma unroll
    for (int i = 0; i < kWirePerWave; ++i) {
        if (k0 + i >= total) break;   // wave-uniform
        const int has0 = bit(m0[i], m1[i], j0), has1 = odd7 ? 0 : bit(m0[i], m1[i], j1);
        const int r0 = below(m0[i], m1[i], j0);
        float2 out;
        out.x = has0 ? values[base + r0] : 0.0f;
        out.y = has1 ? values[base + r0 + has0] : 0.0f;
        reinterpret_cast<float2*>(desc + (k0 + i) * 128)[lane] = out;
        const unsigned short kw = (unsigned short)__shfl((int)w16[i >> 2], 16 * (i & 3) + (lane & 15));
        if (lane < 10) kp16[(k0 + i) * 10 + lane] = kw;
        base += __popcll(m0[i]) + __popcll(m1[i]);
    }
}

size_t wire_blocks(long long total) { return (size_t)((total + kWireBlock - 1) / kWireBlock); }

// d_sums: nb + 1 ints of scratch, d_block_off: nb + 1 long longs of scratch
void launch_wire_unpack(hipStream_t s, const uint8_t* d_records, const float* d_values, long long total, int* d_sums,
                        long long* d_block_off, sift_hip_keypoint* d_kp, float* d_desc) {
    const size_t nb = wire_blocks(total);
    if (!nb) return;
    hipLaunchKernelGGL(wire_mask_count_kernel, dim3((unsigned)nb), dim3(256), 0, s, d_records, total, d_sums);
    hipLaunchKernelGGL(wire_scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)d_sums, (int)nb, d_block_off);
    hipLaunchKernelGGL(wire_unpack_kernel, dim3((unsigned)nb), dim3(256), 0, s, d_records, d_values, total, (const long long*)d_block_off,
                       d_kp, d_desc);
}

// d_sums: [0] the "bin 7 is not +0.0f somewhere" flag, [1 + b] floats on the wire of block b (64 output slots).  counted: the
// descriptor kernel has already filled it for this batch (option "wire_count": kernels_desc.hip), only the scan remains.
void launch_wire_count(hipStream_t s, const float* d_desc, long long total, int* d_sums, long long* d_block_off, bool counted) {
    const size_t nb = wire_blocks(total);
    if (!counted) {
        launch_zero_ints(s, d_sums, 1);
        if (nb) hipLaunchKernelGGL(wire_count_kernel, dim3((unsigned)nb), dim3(256), 0, s, d_desc, total, d_sums + 1, d_sums);
    }
    hipLaunchKernelGGL(wire_scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)(d_sums + 1), (int)nb, d_block_off);
}

void launch_wire_emit(hipStream_t s, const sift_hip_keypoint* d_kp, const float* d_desc, long long total,
                      const long long* d_block_off, uint8_t* d_records, float* d_values) {
    const size_t nb = wire_blocks(total);
    if (nb)
        hipLaunchKernelGGL(wire_emit_kernel, dim3((unsigned)nb), dim3(256), 0, s, d_kp, d_desc, total, d_block_off, d_records,
                           d_values);
}

// The runtime builds a translation unit's device code on the first launch of any of its kernels, and two host threads that make
// their first launches at the same time (several contexts, one thread each) were seen to crash inside that step
// (tools/asan_example.sh: SEGV below hipLaunchKernel).  sift_hip_create touches every unit once, under a lock.
__global__ void tu_probe_wire_kernel() {}
void tu_touch_wire(hipStream_t s) { hipLaunchKernelGGL(tu_probe_wire_kernel, dim3(1), dim3(1), 0, s); }

}  // namespace sift_hip
