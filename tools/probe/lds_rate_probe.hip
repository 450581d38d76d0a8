// Throughput of the LDS operations a tile-resident descriptor stage could build its pixel chains from, in the access shape
// of kernels_desc.hip's update frame (per wave-instruction: 4 rows x 16 consecutive floats of a tile with a row stride of
// 47 floats), with 8 one-wave workgroups per CU as in that kernel:
//   mode 0: 8 x ds_add_f32 (no return) per update
//   mode 1: 8 x ds_read_b32, 8 x v_add_f32, 8 x ds_write_b32 per update (waits for the reads)
//   mode 2: as 1 with ds_read2_b32 / ds_write2_b32 pairs
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/probe/lds_rate_probe.hip -o tools/probe/lds_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int S = 47, EH = 47, MAG = EH * S;

template <int MODE>
__global__ __launch_bounds__(64) void probe(float* out, int iters) {
    __shared__ float s_tile[2 * MAG + 640];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2 * MAG; i += 64) s_tile[i] = (float)i;
    __syncthreads();
    const int baseU = (lane >> 4) * S + (lane & 15);
    const float th = 177.5f + lane, w0 = 0.1f, w1 = 0.2f, w2 = 0.3f, w3 = 0.4f;
    int wx = 3, wy = 5;
    for (int it = 0; it < iters; ++it) {
        float* p = s_tile + wy * S + wx + baseU;
        if (MODE == 0) {
            __hip_atomic_fetch_add(p, th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(p + MAG, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(p + 4 * S, th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(p + 4 * S + MAG, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(p + 8 * S, th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(p + 8 * S + MAG, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(p + 12 * S, th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(p + 12 * S + MAG, w3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
            float o0 = p[0], o1 = p[4 * S], o2 = p[8 * S], o3 = p[12 * S];
            float m0 = p[MAG], m1 = p[4 * S + MAG], m2 = p[8 * S + MAG], m3 = p[12 * S + MAG];
            o0 += th; o1 += th; o2 += th; o3 += th;
            m0 += w0; m1 += w1; m2 += w2; m3 += w3;
            p[0] = o0; p[4 * S] = o1; p[8 * S] = o2; p[12 * S] = o3;
            p[MAG] = m0; p[4 * S + MAG] = m1; p[8 * S + MAG] = m2; p[12 * S + MAG] = m3;
        }
        asm volatile("" ::: "memory");
        wx = (wx + 7) & 31;   // wave-uniform walk over the tile
        wy = (wy + 11) & 31;
    }
    __syncthreads();
    float acc = 0.0f;
    for (int i = lane; i < 2 * MAG; i += 64) acc += s_tile[i];
    out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE>
static void run(const char* what, float* d_out, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<MODE>, dim3(2048), dim3(64), 0, 0, d_out, 16);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(probe<MODE>, dim3(2048), dim3(64), 0, 0, d_out, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // per CU: 8 waves x iters updates
    const double cyc_per_update_per_cu = ms * 1e-3 * 2.4e9 / (8.0 * iters);
    printf("%-40s %8.3f ms for %d updates per wave: %.1f cycles of a CU per update (8 waves per CU, 2.4 GHz assumed)\n", what, ms, iters,
           cyc_per_update_per_cu);
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 2048 * 64 * sizeof(float));
    const int iters = 20000;
    run<0>("8 x ds_add_f32", d_out, iters);
    run<1>("8 x (ds_read, v_add, ds_write)", d_out, iters);
    run<0>("8 x ds_add_f32", d_out, iters);
    run<1>("8 x (ds_read, v_add, ds_write)", d_out, iters);
    return 0;
}
