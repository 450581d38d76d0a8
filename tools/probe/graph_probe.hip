// A chain of dependent launches of the single frame's size (20 launches of ~10 us, each reading what the one before wrote):
// launched one by one on a stream against the same chain captured once into a hipGraph and replayed.  What the gaps between
// dependent dispatches cost on this chip, and whether a graph shortens them (task 7).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/graph_probe.hip -o tools/probe/graph_probe && ./tools/probe/graph_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

__global__ void stage_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int taps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.0f;
    for (int k = -taps; k <= taps; ++k) {
        int j = i + k;
        j = j < 0 ? 0 : (j >= n ? n - 1 : j);
        acc += in[j] * (1.0f / (float)(2 * taps + 1));
    }
    out[i] = acc;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int launches = argc > 1 ? std::atoi(argv[1]) : 20;
    const int n = argc > 2 ? std::atoi(argv[2]) : 1920 * 1080;
    const int taps = 10, reps = 200;
    float *a, *b;
    CK(hipMalloc(&a, (size_t)n * 4));
    CK(hipMalloc(&b, (size_t)n * 4));
    CK(hipMemset(a, 0, (size_t)n * 4));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto chain = [&]() {
        for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(stage_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (l & 1) ? b : a, (l & 1) ? a : b, n, taps);
    };
    // one kernel alone
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    chain();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(stage_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, n, taps);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float one_ms = 0;
    CK(hipEventElapsedTime(&one_ms, e0, e1));
    // the chain, launch by launch, host clock from first launch to completion
    std::vector<double> t_stream, t_graph, t_graph_submit, t_stream_submit;
    for (int r = 0; r < reps; ++r) {
        const double t0 = now_us();
        chain();
        const double t1 = now_us();
        CK(hipStreamSynchronize(s));
        t_stream.push_back(now_us() - t0);
        t_stream_submit.push_back(t1 - t0);
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    for (int r = 0; r < reps; ++r) {
        const double t0 = now_us();
        CK(hipGraphLaunch(ge, s));
        const double t1 = now_us();
        CK(hipStreamSynchronize(s));
        t_graph.push_back(now_us() - t0);
        t_graph_submit.push_back(t1 - t0);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto lo = [](std::vector<double> v) { return *std::min_element(v.begin(), v.end()); };
    std::printf("%d dependent launches over %d floats; one launch alone (events around it): %.1f us\n", launches, n, one_ms * 1000.0f);
    std::printf("stream launches: median %.1f us (min %.1f) submit -> complete, %.1f us on the host to submit; per launch %.1f us\n", med(t_stream), lo(t_stream),
                med(t_stream_submit), med(t_stream) / launches);
    std::printf("graph replay:    median %.1f us (min %.1f) submit -> complete, %.1f us on the host to submit; per launch %.1f us\n", med(t_graph), lo(t_graph),
                med(t_graph_submit), med(t_graph) / launches);
    return 0;
}
