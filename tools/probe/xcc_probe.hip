// Which XCD does workgroup i of a launch run on?  The kernels' XCD-aware orders (blur tiles, extremum-scan tiles, descriptor units)
// assume blockIdx % 8; this prints how often that holds, for the runtime's launch and for hipExtModuleLaunchKernel (the
// library's launch path, launch_cache.h).   hipcc --offload-arch=gfx950 -O2 -o xcc_probe xcc_probe.hip && ./xcc_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void xcc_kernel(int* out) {
    if (threadIdx.x == 0) {
        int id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[blockIdx.x] = id & 0xf;
    }
}

static void report(const char* what, const std::vector<int>& h) {
    int same = 0, hist[16] = {0};
    for (size_t i = 0; i < h.size(); ++i) {
        same += h[i] == (int)(i & 7);
        hist[h[i] & 15]++;
    }
    printf("%-28s %d workgroups: xcc == blockIdx %% 8 for %d; per xcc:", what, (int)h.size(), same);
    for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
    printf("\n");
}

int main() {
    const int n = 4096;
    int* d;
    hipMalloc(&d, n * sizeof(int));
    std::vector<int> h(n);
    hipMemset(d, 0xff, n * sizeof(int));
    xcc_kernel<<<n, 256>>>(d);
    hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
    report("hipLaunchKernel", h);
    hipFunction_t f;
    if (hipGetFuncBySymbol(&f, reinterpret_cast<const void*>(&xcc_kernel)) == hipSuccess) {
        hipMemset(d, 0xff, n * sizeof(int));
        void* args[] = {&d};
        hipExtModuleLaunchKernel(f, n * 256, 1, 1, 256, 1, 1, 0, nullptr, args, nullptr, nullptr, nullptr, 0);
        hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
        report("hipExtModuleLaunchKernel", h);
    }
    hipFree(d);
    return 0;
}
