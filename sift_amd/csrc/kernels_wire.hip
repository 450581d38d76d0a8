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
// contiguous bytes.
#include "common.h"

namespace sift_hip {

constexpr int kWireBlock = 64;   // keypoints per workgroup: 4 waves x 16
constexpr int kWirePerWave = kWireBlock / 4;

__device__ __forceinline__ void wire_presence(const float2 v, int lane, bool& b0, bool& b1) {
    // positions 2*lane and 2*lane+1; bin = position % 8; bin 7 never carries information
    b0 = __float_as_uint(v.x) != 0u;                                  // 2*lane is even: its bin is never 7
    b1 = (((2 * lane + 1) & 7) != 7) && __float_as_uint(v.y) != 0u;
}

// block_sums[gridDim.x] (cleared by the launcher) is raised when some descriptor carries anything but +0.0f in bin 7 of a
// cell: normalizeVector divides every bin by the cell's sum (algorithms.cpp:210-223), so a negative or NaN sum (negative
// pixels, inf * 0 after the cumulative `magnitudes += weighting`) turns the never-written bin into -0.0f or NaN, which
// this format cannot carry
__global__ __launch_bounds__(256) void wire_count_kernel(const float* __restrict__ desc, long long total,
                                                        int* __restrict__ block_sums) {
    __shared__ int s_sum[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int sum = 0;
#pragma unroll 4
    for (int i = 0; i < kWirePerWave; ++i) {
        const long long k = (long long)blockIdx.x * kWireBlock + wave * kWirePerWave + i;
        if (k < total) {
            const float2 v = reinterpret_cast<const float2*>(desc + k * 128)[lane];
            bool b0, b1;
            wire_presence(v, lane, b0, b1);
            sum += __popcll(__ballot(b0)) + __popcll(__ballot(b1));
            if (__ballot((lane & 3) == 3 && __float_as_uint(v.y) != 0u) != 0ull && lane == 0) atomicOr(&block_sums[gridDim.x], 1);
        }
    }
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// exclusive scan of the block sums (one workgroup); block_off[nb] = total number of floats
__global__ __launch_bounds__(1024) void wire_scan_kernel(const int* __restrict__ block_sums, int nb,
                                                         long long* __restrict__ block_off) {
    __shared__ long long s_part[1024];
    const int tid = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, nb);
    long long sum = 0;
    for (int i = lo; i < hi; ++i) sum += block_sums[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const long long v = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    long long run = s_part[tid] - sum;
    for (int i = lo; i < hi; ++i) {
        block_off[i] = run;
        run += block_sums[i];
    }
    if (tid == 1023) block_off[nb] = s_part[1023];
}

__global__ __launch_bounds__(256) void wire_emit_kernel(const sift_hip_keypoint* __restrict__ kp,
                                                       const float* __restrict__ desc, long long total,
                                                       const long long* __restrict__ block_off,
                                                       uint8_t* __restrict__ records, float* __restrict__ values) {
    __shared__ int s_sum[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    float2 v[kWirePerWave];
    unsigned long long even[kWirePerWave], odd[kWirePerWave];
    int sum = 0;
#pragma unroll
    for (int i = 0; i < kWirePerWave; ++i) {
        const long long k = (long long)blockIdx.x * kWireBlock + wave * kWirePerWave + i;
        even[i] = odd[i] = 0ull;
        v[i] = make_float2(0.0f, 0.0f);
        if (k < total) {
            v[i] = reinterpret_cast<const float2*>(desc + k * 128)[lane];
            bool b0, b1;
            wire_presence(v[i], lane, b0, b1);
            even[i] = __ballot(b0);
            odd[i] = __ballot(b1);
            sum += __popcll(even[i]) + __popcll(odd[i]);
        }
    }
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    long long base = block_off[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += s_sum[w];
    const uint8_t* __restrict__ kpb = reinterpret_cast<const uint8_t*>(kp);
#pragma unroll
    for (int i = 0; i < kWirePerWave; ++i) {
        const long long k = (long long)blockIdx.x * kWireBlock + wave * kWirePerWave + i;
        if (k < total) {
            // values, ascending position: lane's even element, then its odd one
            const int r = __popcll(even[i] & below) + __popcll(odd[i] & below);
            const bool b0 = (even[i] >> lane) & 1ull, b1 = (odd[i] >> lane) & 1ull;
            if (b0) values[base + r] = v[i].x;
            if (b1) values[base + r + (b0 ? 1 : 0)] = v[i].y;
            base += __popcll(even[i]) + __popcll(odd[i]);
            // record: 20 bytes of the keypoint, 14 bytes of presence bits
            if (lane < 34) {
                unsigned byte;
                if (lane < 20) {
                    byte = kpb[k * 20 + lane];
                } else {
                    byte = 0u;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int j = 8 * (lane - 20) + t;          // < 112
                        const int p = (j / 7) * 8 + (j % 7);        // position among the 128 floats
                        const unsigned long long m = (p & 1) ? odd[i] : even[i];
                        byte |= (unsigned)((m >> (p >> 1)) & 1ull) << t;
                    }
                }
                records[k * 34 + lane] = (uint8_t)byte;
            }
        }
    }
}

size_t wire_blocks(long long total) { return (size_t)((total + kWireBlock - 1) / kWireBlock); }

void launch_wire_count(hipStream_t s, const float* d_desc, long long total, int* d_block_sums, long long* d_block_off) {
    const size_t nb = wire_blocks(total);
    (void)hipMemsetAsync(d_block_sums + nb, 0, sizeof(int), s);   // "bin 7 is not +0.0f somewhere"
    if (nb) hipLaunchKernelGGL(wire_count_kernel, dim3((unsigned)nb), dim3(256), 0, s, d_desc, total, d_block_sums);
    hipLaunchKernelGGL(wire_scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)d_block_sums, (int)nb, d_block_off);
}

void launch_wire_emit(hipStream_t s, const sift_hip_keypoint* d_kp, const float* d_desc, long long total,
                      const long long* d_block_off, uint8_t* d_records, float* d_values) {
    const size_t nb = wire_blocks(total);
    if (nb)
        hipLaunchKernelGGL(wire_emit_kernel, dim3((unsigned)nb), dim3(256), 0, s, d_kp, d_desc, total, d_block_off, d_records,
                           d_values);
}

}  // namespace sift_hip
