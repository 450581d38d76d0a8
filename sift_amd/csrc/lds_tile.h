// LDS helpers shared by the tile kernels (kernels_pyramid.hip, kernels_chain.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sift_hip {

constexpr int gcd_ce(int a, int b) { return b == 0 ? a : gcd_ce(b, a % b); }

// 16-byte LDS reads that the optimiser cannot split into narrower (slower, conflicting) reads of just
// the elements it can prove are used.  Reads and their s_waitcnt live in ONE asm statement, so no
// output register can be touched before the data has landed.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void lds_read_b128x4(const float4* p, float4& a, float4& b, float4& c, float4& d) {
    asm volatile(
        "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
        "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
        : "v"(lds_addr(p))
        : "memory");
}
__device__ __forceinline__ void lds_read_b128x2(const float4* p, float4& a, float4& b) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(lds_addr(p))
                 : "memory");
}
__device__ __forceinline__ void lds_read_b128x1(const float4* p, float4& a) {
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a) : "v"(lds_addr(p)) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_read_window(const float4* p, float4 (&f)[N]) {
    constexpr int N4 = N / 4 * 4;
#pragma unroll
    for (int c = 0; c < N4; c += 4) lds_read_b128x4(p + c, f[c], f[c + 1], f[c + 2], f[c + 3]);
    if constexpr (N - N4 >= 2) lds_read_b128x2(p + N4, f[N4], f[N4 + 1]);
    if constexpr ((N - N4) & 1) lds_read_b128x1(p + N - 1, f[N - 1]);
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace sift_hip
