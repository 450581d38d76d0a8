// What this box's HBM delivers to plain streaming kernels - the practical ceiling next to the 8 TB/s of the data sheet that
// bench.py's roofline.frac is priced against.  Buffers of 1 GB (far beyond the 256 MB Infinity Cache), 16 bytes per lane and
// access, grid-stride, 8 launches each:
//   read   : every byte read once (sum kept in registers)            bytes = N
//   write  : every byte written once                                  bytes = N
//   copy   : read N, write N (the blur kernels' shape: 4 B in, 4 B out per pixel)      bytes = 2 N
//   copy12 : read N, write 2 N (blur + DoG: 4 B in, 8 B out)                            bytes = 3 N
//   hipcc --offload-arch=gfx950 -O3 tools/probe/hbm_copy_probe.hip -o tools/probe/hbm_copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ a, f4* __restrict__ b, f4* __restrict__ c, size_t n16, float* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        if (MODE == 0) acc += a[i];
        if (MODE == 1) b[i] = f4{1.f, 2.f, 3.f, (float)i};
        if (MODE == 2) b[i] = a[i];
        if (MODE == 3) { const f4 v = a[i]; b[i] = v; c[i] = v * 2.0f; }
    }
    if (MODE == 0 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <int MODE>
static void run(const char* what, double bytes_per_n, f4* a, f4* b, f4* c, size_t n16, float* sink, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, a, b, c, n16, sink);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, a, b, c, n16, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double gb = bytes_per_n * (double)n16 * 16.0 * 8 / 1e9;
    printf("%-8s grid %5d: %7.1f GB/s  (%.3f ms per launch)\n", what, grid, gb / (ms / 1e3), ms / 8);
}

int main() {
    const size_t n16 = (size_t)1 << 26;   // 1 GiB per buffer
    f4 *a, *b, *c;
    float* sink;
    hipMalloc(&a, n16 * 16); hipMalloc(&b, n16 * 16); hipMalloc(&c, n16 * 16); hipMalloc(&sink, 4);
    hipMemset(a, 1, n16 * 16); hipMemset(b, 0, n16 * 16); hipMemset(c, 0, n16 * 16);
    for (int grid : {2048, 8192}) {
        run<0>("read", 1, a, b, c, n16, sink, grid);
        run<1>("write", 1, a, b, c, n16, sink, grid);
        run<2>("copy", 2, a, b, c, n16, sink, grid);
        run<3>("copy12", 3, a, b, c, n16, sink, grid);
    }
    return 0;
}
