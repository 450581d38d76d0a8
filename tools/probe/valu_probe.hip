// Micro-benchmark: issue rate of scalar vs packed FP32 VALU ops on gfx950 (per SIMD, per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int MODE>
__global__ void probe(float* out, int iters, float s0) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    f2 sv = {s0, s0};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 8 independent v_mul_f32 per group
            REP64(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                               "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0));)
        } else if (MODE == 1) {  // 8 independent v_pk_mul_f32
            REP64(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                               "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(sv));)
        } else if (MODE == 2) {  // dependent chain v_pk_add_f32
            REP64(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n"
                               "v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1"
                               : "+v"(p0) : "v"(sv));)
        } else if (MODE == 3) {  // dependent chain v_add_f32
            REP64(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                               "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1"
                               : "+v"(a0) : "v"(s0));)
        } else if (MODE == 4) {  // 8 independent v_pk_add_f32
            REP64(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                               "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(sv));)
        } else if (MODE == 5) {  // 8 independent v_pk_fma_f32
            REP64(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                               "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(sv));)
        } else if (MODE == 6) {  // 8 independent v_fma_f32
            REP64(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                               "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0));)
        } else if (MODE == 7) {  // pk_mul with SGPR pair source
            REP64(asm volatile("v_pk_mul_f32 %0, %0, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %8 op_sel_hi:[1,0]\n"
                               "v_pk_mul_f32 %4, %4, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %5, %5, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %6, %6, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %7, %8 op_sel_hi:[1,0]"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(sv));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x + p4.y + p5.y + p6.y + p7.y;
}

template <int MODE>
void run(const char* name, int waves_per_simd, float* d) {
    const int iters = 200;
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 64 * 8;
    const double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
    printf("%-34s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (%.2f clk @2.4GHz)\n", name, waves_per_simd, ms,
           ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
    for (int wps : {1, 2, 4}) {
        run<0>("v_mul_f32 x8 independent", wps, d);
        run<1>("v_pk_mul_f32 x8 independent", wps, d);
        run<4>("v_pk_add_f32 x8 independent", wps, d);
        run<7>("v_pk_mul_f32 x8 indep, SGPR src", wps, d);
        run<5>("v_pk_fma_f32 x8 independent", wps, d);
        run<6>("v_fma_f32 x8 independent", wps, d);
        run<2>("v_pk_add_f32 dependent chain", wps, d);
        run<3>("v_add_f32 dependent chain", wps, d);
    }
    return 0;
}
