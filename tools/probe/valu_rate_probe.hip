// What the vector ALU sustains for the blurs' arithmetic: plain v_mul_f32 / v_add_f32 against v_pk_mul_f32 / v_pk_add_f32, at 1, 2, 3, 4
// and 8 waves per SIMD.     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe
// Prints lane-operations per second (one multiply or one add of one lane = 1) and cycles per instruction and SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: scalar mul + add, 1: packed mul + add; + 2: the add right behind its multiply (dependent neighbours)
__global__ __launch_bounds__(256) void valu_kernel(float* out, int iters, float seed) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)i + (float)threadIdx.x;
    const float m = 1.0000001f, c = 1e-7f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // 16 independent multiplies, then the 16 adds: a dependent instruction is 16 issues behind
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 1) {
            f2 v[8];
            const f2 mm = {m, m}, cc = {c, c};
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = f2{a[2 * i], a[2 * i + 1]};
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(v[i]) : "v"(mm));
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(v[i]) : "v"(cc));
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[2 * i] = v[i].x; a[2 * i + 1] = v[i].y; }
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f2 v = {a[i], a[i + 1]};
                f2 mm = {m, m}, cc = {c, c};
                asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(v) : "v"(mm));
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(v) : "v"(cc));
                a[i] = v.x;
                a[i + 1] = v.y;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char* what, int waves_per_simd) {
    float* d;
    hipMalloc(&d, 4);
    const int iters = 20000;
    const int grid = 256 * waves_per_simd;   // 256-thread workgroups: one wave per SIMD each
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    valu_kernel<MODE><<<grid, 256>>>(d, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    valu_kernel<MODE><<<grid, 256>>>(d, iters, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double lane_ops = (double)grid * 256.0 * iters * 32.0;              // 16 values x (mul + add)
    const double instr_per_simd = (double)waves_per_simd * iters * ((MODE & 1) ? 16.0 : 32.0);
    printf("%-28s %d waves/SIMD: %7.3f ms  %6.2f T lane-ops/s  %5.2f cycles per instruction and SIMD at 2.4 GHz\n", what, waves_per_simd, ms,
           lane_ops / ms / 1e9, ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 3, 4, 8}) {
        run<0>("mul, add; independent", w);
        run<1>("pk_mul, pk_add; independent", w);
        run<2>("mul, add; dependent pairs", w);
        run<3>("pk_mul, pk_add; dependent", w);
    }
    return 0;
}
