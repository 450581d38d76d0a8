// What this box's HBM delivers to streaming kernels - the practical ceiling next to the 8 TB/s of the data sheet that
// bench.py's roofline.frac is priced against (MI355X_MICROARCH.md records 6.29 TB/s for a float4 copy).  Buffers of 1 GiB
// (far beyond the 256 MB Infinity Cache), 16 bytes per lane and access, 10 launches each after a warm-up launch.
//   read   : every byte read once (sum kept in registers)                               bytes = N
//   write  : every byte written once                                                    bytes = N
//   copy   : read N, write N (the blur kernels' shape: 4 B in, 4 B out per pixel)       bytes = 2 N
//   copy12 : read N, write 2 N (blur + DoG: 4 B in, 8 B out)                            bytes = 3 N
// Round 5: round 4's probe was a naive grid-stride loop (ONE 16-byte access in flight per lane, ordinary stores) and reported
// 4.7 - 4.9 TB/s for a copy.  This one sweeps what a streaming kernel can choose: U independent 16-byte loads in flight per
// lane (each wave takes U consecutive 1 KiB pieces, all loads issued before the first store), non-temporal loads and / or
// stores, and the number of workgroups per CU (persistent grid-stride over pieces).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/hbm_copy_probe.hip -o tools/probe/hbm_copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0 read, 1 write, 2 copy, 3 copy12.  U loads in flight per lane; NTL / NTS: non-temporal loads / stores.
template <int MODE, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ a, f4* __restrict__ b, f4* __restrict__ c, size_t n16, float* sink) {
    const size_t lane = threadIdx.x & 63, wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t waves = (size_t)gridDim.x * 4;
    const size_t chunk = (size_t)64 * U;   // 16-byte units a wave moves per trip
    f4 acc = {0, 0, 0, 0};
    for (size_t base = wave * chunk; base + chunk <= n16; base += waves * chunk) {
        f4 v[U];
        if (MODE != 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(a + base + (size_t)u * 64 + lane) : a[base + (size_t)u * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 64 + lane;
            if (MODE == 0) acc += v[u];
            if (MODE == 1) { const f4 w = f4{1.f, 2.f, 3.f, (float)i}; if (NTS) __builtin_nontemporal_store(w, b + i); else b[i] = w; }
            if (MODE == 2) { if (NTS) __builtin_nontemporal_store(v[u], b + i); else b[i] = v[u]; }
            if (MODE == 3) {
                if (NTS) { __builtin_nontemporal_store(v[u], b + i); __builtin_nontemporal_store(v[u] * 2.0f, c + i); }
                else { b[i] = v[u]; c[i] = v[u] * 2.0f; }
            }
        }
    }
    if (MODE == 0 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <int MODE, int U, bool NTL, bool NTS>
static double run(f4* a, f4* b, f4* c, size_t n16, float* sink, int grid) {
    static const double bytes_per_n[4] = {1, 1, 2, 3};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, U, NTL, NTS>), dim3(grid), dim3(256), 0, 0, a, b, c, n16, sink);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<MODE, U, NTL, NTS>), dim3(grid), dim3(256), 0, 0, a, b, c, n16, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return bytes_per_n[MODE] * (double)n16 * 16.0 * 10 / 1e9 / (ms / 1e3);
}

template <int U, bool NTL, bool NTS>
static void sweep(f4* a, f4* b, f4* c, size_t n16, float* sink) {
    for (int per_cu : {2, 4, 8}) {   // workgroups of 4 waves per CU: 8, 16, 32 waves per CU
        const int grid = 256 * per_cu;
        printf("U=%d loads in flight, nt loads %d, nt stores %d, %2d waves/CU:  read %7.1f  write %7.1f  copy %7.1f  copy12 %7.1f  GB/s\n", U, (int)NTL, (int)NTS,
               4 * per_cu, run<0, U, NTL, NTS>(a, b, c, n16, sink, grid), run<1, U, NTL, NTS>(a, b, c, n16, sink, grid),
               run<2, U, NTL, NTS>(a, b, c, n16, sink, grid), run<3, U, NTL, NTS>(a, b, c, n16, sink, grid));
    }
}

int main() {
    const size_t n16 = (size_t)1 << 26;   // 1 GiB per buffer
    f4 *a, *b, *c;
    float* sink;
    hipMalloc(&a, n16 * 16); hipMalloc(&b, n16 * 16); hipMalloc(&c, n16 * 16); hipMalloc(&sink, 4);
    hipMemset(a, 1, n16 * 16); hipMemset(b, 0, n16 * 16); hipMemset(c, 0, n16 * 16);
    sweep<1, false, false>(a, b, c, n16, sink);
    sweep<4, false, false>(a, b, c, n16, sink);
    sweep<4, false, true>(a, b, c, n16, sink);
    sweep<4, true, true>(a, b, c, n16, sink);
    sweep<8, false, true>(a, b, c, n16, sink);
    sweep<8, true, true>(a, b, c, n16, sink);
    return 0;
}
