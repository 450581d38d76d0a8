// Plumbing kernels: device-to-device copies as a kernel of the library (the one-process multi-GPU gather, group.cpp) and the few
// counters cleared between kernels of a stream.  Transfers between page-locked host memory and HBM as kernels of our own (options "io_kernels",
// "stage_kernels") were measured in rounds 2 - 3, lost to the runtime's copies (6.5 - 6.8 ms a step against
// 4.5 - 5.5 ms; reads over the link reach ~28 GB/s with this much in flight) and were removed in round 4.
#include <algorithm>

#include <atomic>
#include <chrono>
#include <unordered_map>
#include <utility>
#include <vector>

#include "common.h"

namespace sift_hip {

// (the launch locks, the thread's tracked device and the table of cached function objects are host code: launch_guard.cpp)

constexpr int kIoWorkgroups = 16;
constexpr int kIoUnroll = 4;

typedef unsigned u4v __attribute__((ext_vector_type(4)));   // a 16-byte unit the nontemporal builtins accept
typedef float f4v __attribute__((ext_vector_type(4)));

// 16-byte units [0, n16) of src -> dst (device memory of this GPU, or a peer's through peer access)
__global__ __launch_bounds__(256) void io_copy_kernel(const u4v* __restrict__ src, u4v* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (kIoUnroll - 1) * stride < n16; i += kIoUnroll * stride) {
        u4v v[kIoUnroll];
#pragma unroll
        for (int k = 0; k < kIoUnroll; ++k) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < kIoUnroll; ++k) __builtin_nontemporal_store(v[k], dst + i + k * stride);
    }
    for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

__global__ void io_tail_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int n) {
    const int i = threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void io_copy4_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// bytes [0, bytes) of src -> dst; both 16-byte aligned (4-byte aligned pointers and sizes take a narrower kernel)
void launch_io_copy(hipStream_t s, const void* src, void* dst, size_t bytes) {
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) != 0 && bytes % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3u) == 0) {
        if (bytes) hipLaunchKernelGGL(io_copy4_kernel, dim3(4 * kIoWorkgroups), dim3(256), 0, s, static_cast<const unsigned*>(src), static_cast<unsigned*>(dst), bytes / 4);
        return;
    }
    const size_t n16 = bytes / 16;
    if (n16) hipLaunchKernelGGL(io_copy_kernel, dim3(kIoWorkgroups), dim3(256), 0, s, static_cast<const u4v*>(src), static_cast<u4v*>(dst), n16);
    const int tail = (int)(bytes - n16 * 16);
    if (tail)
        hipLaunchKernelGGL(io_tail_kernel, dim3(1), dim3(64), 0, s, static_cast<const unsigned char*>(src) + n16 * 16,
                           static_cast<unsigned char*>(dst) + n16 * 16, tail);
}

// A few counters to zero between kernels of one stream: a kernel of our own, not hipMemsetAsync - the runtime's fill kernel was
// measured at 100 - 230 us for 128 bytes whenever bandwidth-bound kernels were running beside it (profiles/r02_timeline_*:
// the gradient pass waited that long behind it), an ordinary one-workgroup kernel takes a few microseconds.
__global__ void zero_ints_kernel(int* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0;
}

void launch_zero_ints(hipStream_t s, int* p, size_t n) {
    if (n == 0) return;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 64);
    hipLaunchKernelGGL(zero_ints_kernel, dim3(grid), dim3(256), 0, s, p, n);
}

// The runtime builds a translation unit's device code on the first launch of any of its kernels, and two host threads that make
// their first launches at the same time (several contexts, one thread each) were seen to crash inside that step
// (tools/asan_example.sh: SEGV below hipLaunchKernel).  sift_hip_create touches every unit once, under a lock.
__global__ void tu_probe_io_kernel() {}
void tu_touch_io(hipStream_t s) { hipLaunchKernelGGL(tu_probe_io_kernel, dim3(1), dim3(1), 0, s); }

}  // namespace sift_hip

// device-to-device copy as a kernel of this library (group.cpp: the gather of shards that share a GPU, or that reach the first
// GPU through peer access); both pointers 16-byte aligned
extern "C" int sift_hip_internal_copy(void* stream, const void* src, void* dst, size_t bytes) {
    sift_hip::launch_io_copy(static_cast<hipStream_t>(stream), src, dst, bytes);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
