// The pyramid's streaming blur kernels ALONE on the chip, on the bench's launch shapes (32 frames of 1920x1080 and of the
// next octave, radii 7 / 10 / 14): blur_stream_kernel (rounds 1 - 5: adjacent columns packed) against blur_stream2_kernel
// (round 6: consecutive rows packed) at several cuts of the launch, output compared bit for bit.  Includes the kernels'
// own header, so what it times is what the library launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe/blur_probe.hip -o tools/probe/blur_probe && ./tools/probe/blur_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../sift_amd/csrc/blur_stream.h"

using namespace sift_hip;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

struct Cut { int strips, strip_w, chunks, chunk_h, total; };

static Cut cut_for(int w, int h, int n, int radius, int target) {
    const int RI = stream_runin(radius);
    Cut c;
    c.strips = (w + 127) / 128;
    c.strip_w = (((w + c.strips - 1) / c.strips) + 1) / 2 * 2;
    int chunks = target / (n * c.strips);
    if (chunks < 1) chunks = 1;
    int chunk_h = (h + chunks - 1) / chunks;
    if (chunk_h < 3 * RI) chunk_h = 3 * RI;
    chunk_h = (chunk_h + 3) / 4 * 4;
    c.chunk_h = chunk_h;
    c.chunks = (h + chunk_h - 1) / chunk_h;
    c.total = n * c.strips * c.chunks;
    return c;
}

template <class F>
static float time_launches(F&& launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, nullptr));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b, nullptr));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms * 1000.0f / reps;
}

template <int R>
static void run_radius(const float* d_in, float* d_ref, float* d_out, int w, int h, int n) {
    // taps: Kernel1D::initGaussian's shape for the sigma whose radius is R (symmetric bit for bit, like the library's)
    const float sigma = (float)((R - 0.25) / 3.0);
    std::vector<float> t(2 * R + 1);
    float sum = 0;
    for (int x = -R; x <= R; ++x) { t[x + R] = std::exp(-0.5f * x * x / (sigma * sigma)); }
    for (int i = 0; i <= R; ++i) t[2 * R - i] = t[i];
    for (float v : t) sum += v;
    for (float& v : t) v /= sum;
    for (int i = 0; i <= R; ++i) t[2 * R - i] = t[i];
    float* d_taps;
    CK(hipMalloc(&d_taps, t.size() * sizeof(float)));
    CK(hipMemcpy(d_taps, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
    const size_t px = (size_t)w * h * n;
    const double gb = px * 8.0 / 1e9;
    const StreamDecimate none{nullptr, nullptr, 0, 0, nullptr};
    std::vector<float> ref(px), got(px);

    const Cut c0 = cut_for(w, h, n, R, 2048);
    auto old_launch = [&]() {
        hipLaunchKernelGGL((blur_stream_kernel<R, false, 2, false>), dim3((c0.total + 3) / 4), dim3(256), 0, nullptr, d_in, d_ref, (float*)nullptr, w, h, c0.strips,
                           c0.strip_w, c0.chunks, c0.chunk_h, c0.total, (const float*)d_taps, none);
    };
    CK(hipMemset(d_ref, 0xff, px * sizeof(float)));
    const float us_old = time_launches(old_launch, 20);
    CK(hipGetLastError());
    CK(hipMemcpy(ref.data(), d_ref, px * sizeof(float), hipMemcpyDeviceToHost));
    std::printf("R %2d  %4dx%-4d x%d  columns packed (r05), %5d waves (chunks of %3d rows): %7.1f us  %5.2f TB/s\n", R, w, h, n, c0.total, c0.chunk_h, us_old,
                gb / us_old * 1e3);
    for (int target : {1536, 2048, 2560, 3072, 4096}) {
        const Cut c = cut_for(w, h, n, R, target);
        if (target != 2048 && c.total == cut_for(w, h, n, R, 2048).total) continue;
        for (int var = 0; var < 2; ++var) {
            auto new_launch = [&]() {
                if (var == 0)
                    hipLaunchKernelGGL((blur_stream2_kernel<R, 0>), dim3((c.total + 3) / 4), dim3(256), 0, nullptr, d_in, d_out, w, h, c.strips, c.strip_w, c.chunks,
                                       c.chunk_h, c.total, (const float*)d_taps);
                else
                    hipLaunchKernelGGL((blur_stream2_kernel<R, 1>), dim3((c.total + 3) / 4), dim3(256), 0, nullptr, d_in, d_out, w, h, c.strips, c.strip_w, c.chunks,
                                       c.chunk_h, c.total, (const float*)d_taps);
            };
            CK(hipMemset(d_out, 0xee, px * sizeof(float)));
            const float us = time_launches(new_launch, 20);
            CK(hipGetLastError());
            CK(hipMemcpy(got.data(), d_out, px * sizeof(float), hipMemcpyDeviceToHost));
            const bool same = std::memcmp(got.data(), ref.data(), px * sizeof(float)) == 0;
            size_t first = 0;
            if (!same)
                for (size_t i = 0; i < px; ++i)
                    if (std::memcmp(&got[i], &ref[i], 4) != 0) { first = i; break; }
            std::printf("R %2d  %4dx%-4d x%d  rows packed (var %d),   %5d waves (chunks of %3d rows): %7.1f us  %5.2f TB/s  %s", R, w, h, n, var, c.total, c.chunk_h, us,
                        gb / us * 1e3, same ? "bit-identical\n" : "DIFFERS");
            if (!same)
                std::printf(" first at image %zu row %zu column %zu: %g vs %g\n", first / ((size_t)w * h), first % ((size_t)w * h) / w, first % w, got[first], ref[first]);
        }
    }
    CK(hipFree(d_taps));
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 32;
    const int W = 1920, H = 1080;
    const size_t px = (size_t)W * H * n;
    std::vector<float> host(px);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < px; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        host[i] = (float)((s >> 40) & 255u) + (float)((s >> 20) & 1023u) / 1024.0f;
    }
    float *d_in, *d_ref, *d_out;
    CK(hipMalloc(&d_in, px * sizeof(float)));
    CK(hipMalloc(&d_ref, px * sizeof(float)));
    CK(hipMalloc(&d_out, px * sizeof(float)));
    CK(hipMemcpy(d_in, host.data(), px * sizeof(float), hipMemcpyHostToDevice));
    for (int oct = 0; oct < 2; ++oct) {
        const int w = W >> oct, h = H >> oct;
        run_radius<7>(d_in, d_ref, d_out, w, h, n);
        run_radius<10>(d_in, d_ref, d_out, w, h, n);
        run_radius<14>(d_in, d_ref, d_out, w, h, n);
    }
    // a ragged shape: strips that end inside the image, an odd number of row pairs per chunk, one image
    run_radius<7>(d_in, d_ref, d_out, 1000, 762, 3);
    run_radius<10>(d_in, d_ref, d_out, 1322, 500, 2);
    return 0;
}
