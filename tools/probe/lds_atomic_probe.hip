// Is the LDS float atomic (ds_add_f32) the same operation as v_add_f32 (round to nearest even, denormals kept, the
// same NaN results) and do a wave's consecutive ds_add_f32 to one address apply in program order?  Both are what a
// descriptor stage that keeps its pixel chains in LDS (kernels_desc.hip) would rely on.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/probe/lds_atomic_probe.hip -o tools/probe/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void probe(const float* a, const float* b, const float* c, float* out_lds, float* out_valu, int n) {
    __shared__ float s[256];
    const int tid = threadIdx.x;
    for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
        s[tid] = a[i];
        __builtin_amdgcn_s_waitcnt(0);
        // two consecutive adds to the same word, never waited for in between
        asm volatile("ds_add_f32 %0, %1\n\tds_add_f32 %0, %2" ::"v"((unsigned)(tid * 4)), "v"(b[i]), "v"(c[i]) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        out_lds[i] = s[tid];
        float v = a[i];
        v = v + b[i];
        v = v + c[i];
        out_valu[i] = v;
    }
}

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint32_t next32() {
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    return (uint32_t)(rng >> 16);
}
static float pick(int kind) {
    uint32_t u = next32();
    switch (kind % 6) {
        case 0: break;                                             // any bit pattern (NaNs, infinities included)
        case 1: u = (u & 0x807fffffu);  break;                     // denormal or zero
        case 2: u = (u & 0x807fffffu) | (1u << 23); break;         // smallest normals
        case 3: u = (u & 0x007fffffu) | ((100u + (u >> 28)) << 23); break;   // mid range
        case 4: { float f = (float)(u % 360000u) / 1000.0f; memcpy(&u, &f, 4); } break;   // angles
        case 5: u = (u & 0x807fffffu) | ((126u + (u >> 30)) << 23); break;   // around one: cancellation into denormals is rare, rounding ties common
    }
    float f; memcpy(&f, &u, 4);
    return f;
}

int main() {
    const int n = 1 << 22;
    std::vector<float> a(n), b(n), c(n), ol(n), ov(n);
    for (int i = 0; i < n; ++i) { a[i] = pick(i); b[i] = pick(i / 6); c[i] = pick(i / 36); }
    float *da, *db, *dc, *dl, *dv;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dl, n * 4); hipMalloc(&dv, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, da, db, dc, dl, dv, n);
    hipMemcpy(ol.data(), dl, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ov.data(), dv, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_nonnan = 0, host_bad = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t x, y, z;
        memcpy(&x, &ol[i], 4); memcpy(&y, &ov[i], 4);
        volatile float h = a[i]; h = h + b[i]; h = h + c[i];
        float hh = h; memcpy(&z, &hh, 4);
        const bool isnan_ = ov[i] != ov[i];
        if (x != y) {
            ++bad;
            if (!isnan_) {
                if (bad_nonnan < 10) printf("  a=%a b=%a c=%a lds=%08x valu=%08x\n", a[i], b[i], c[i], x, y);
                ++bad_nonnan;
            }
        }
        if (!isnan_ && y != z) ++host_bad;
    }
    printf("ds_add_f32 vs v_add_f32 over %d triples: %ld differ (%ld of them not NaN); v_add_f32 vs host: %ld differ\n", n, bad,
           bad_nonnan, host_bad);
    return bad_nonnan ? 1 : 0;
}
