// A HIP runtime and a context layer made of plain host code, for ONE purpose: to compile the library's multi-threaded host code -
// sift_amd/csrc/group.cpp (shard threads, gather thread, two batches in flight), phase_gate.h, launch_guard.h's locks - exactly
// as it ships and run it under ThreadSanitizer / AddressSanitizer on the CPU (tests/test_host_tsan.py).  Device
// memory is host memory, streams run what they are given at once, events are always complete; the "contexts" return
// deterministic keypoint lists computed from the frames' first pixels, in the library's real record layout, and pack / unpack
// them in the real sparse wire format (34-byte records + the floats that are set), so the group's gather is checked end to end.
// Test infrastructure only: nothing here is linked into libsift_hip.so.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sift_hip.h"
#include "../../sift_amd/csrc/launch_guard.h"

// ---- launch locks: the shipped implementation is sift_amd/csrc/launch_guard.cpp; the group / gate drivers link this small copy,
// the collect driver (-DSIFT_FAKE_WITH_REAL_LAUNCH_GUARD) links the real file ---------------------------------------------------
namespace sift_hip {
#ifndef SIFT_FAKE_WITH_REAL_LAUNCH_GUARD
std::recursive_mutex& launch_lock_of(int device) {
    static std::recursive_mutex m[64];
    return m[(unsigned)device % 64u];
}
static thread_local int t_device = 0;
int set_device_tracked(int device) { t_device = device; return (int)hipSuccess; }
int tracked_device() { return t_device; }
int current_device_refreshed() { return t_device; }
std::recursive_mutex& launch_lock() { return launch_lock_of(t_device); }
double launch_lock_wait_ms() { return 0.0; }
LaunchGuard::LaunchGuard() : m(launch_lock()) { m.lock(); }
LaunchGuard::LaunchGuard(int device) : m(launch_lock_of(device)) { m.lock(); }
static int& fake_device() { return t_device; }
#else
static int& fake_device() { static thread_local int d = 0; return d; }
#endif
}  // namespace sift_hip

// ---- the runtime ----------------------------------------------------------------------------------------------------------------
static std::atomic<long long> g_live_allocs{0};
extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 4; return hipSuccess; }
hipError_t hipSetDevice(int d) { sift_hip::fake_device() = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = sift_hip::fake_device(); return hipSuccess; }
// a function object per (device, host stub): the address of a cell in a per-device table, filled on first use WITHOUT a lock of its
// own - the real runtime's table is what the library's launch lock protects, and a second thread inside it would be the bug
hipError_t hipGetFuncBySymbol(hipFunction_t* f, const void* stub) {
    static const void* seen[4][64];
    const int d = sift_hip::fake_device() & 3;
    for (int i = 0; i < 64; ++i) {
        if (seen[d][i] == stub || seen[d][i] == nullptr) {
            seen[d][i] = stub;
            *f = reinterpret_cast<hipFunction_t>(&seen[d][i]);
            return hipSuccess;
        }
    }
    return hipErrorInvalidDeviceFunction;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "fake"; }
hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); g_live_allocs++; return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { if (p) { std::free(p); g_live_allocs--; } return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { std::memset(a, 0, sizeof(*a)); return hipErrorInvalidValue; }   // "ordinary memory"
long long fake_live_allocs(void) { return g_live_allocs.load(); }
}

// ---- the context layer ------------------------------------------------------------------------------------------------------
// Keypoints of a frame: n = 1 + (int)pixel0 % 7 records; record j of a frame with first pixel v has x = (int)v + j, y = j,
// scale = v, orientation = 177.5f and a descriptor with floats at positions (j + 3 k) % 128, bin 7 left +0.0f.
// A frame whose first pixel is negative makes the batch fail (status SIFT_HIP_EPRECONDITION for that image).
struct sift_hip_ctx {
    int device = 0;
    std::vector<sift_hip_keypoint> kp;
    std::vector<float> desc;
    std::vector<int32_t> status, counts;
    bool have = false;
    int64_t nnz = 0;
};

static void fake_points(float v, std::vector<sift_hip_keypoint>& kp, std::vector<float>& desc, int count = 0) {
    const int n = count > 0 ? count : 1 + ((int)v % 7 + 7) % 7;
    for (int j = 0; j < n; ++j) {
        sift_hip_keypoint r;
        std::memset(&r, 0, sizeof(r));
        r.scale = v; r.orientation = 177.5f;
        r.x = (uint16_t)((int)v + j); r.y = (uint16_t)j; r.octave = 1; r.index = 1; r.filtered = 0; r.has_descriptor = 1;
        kp.push_back(r);
        const size_t base = desc.size();
        desc.resize(base + 128, 0.0f);
        for (int k = 0; k < 20; ++k) {
            const int p = (j + 3 * k) % 128;
            if ((p & 7) != 7) desc[base + (size_t)p] = v + 0.25f * (float)k;
        }
    }
}

extern "C" {
int sift_hip_create(int device, sift_hip_ctx** out, char*, int) {
    *out = new sift_hip_ctx();
    (*out)->device = device;
    return SIFT_HIP_OK;
}
void sift_hip_destroy(sift_hip_ctx* c) { delete c; }
int sift_hip_set_option(sift_hip_ctx*, const char*, int) { return SIFT_HIP_OK; }
int sift_hip_calculate_batch(sift_hip_ctx* c, const float* imgs, int n, int w, int h, const sift_hip_params*, char* err, int errlen) {
    c->kp.clear(); c->desc.clear(); c->status.assign((size_t)n, 0); c->counts.assign((size_t)n, 0);
    int rc = SIFT_HIP_OK;
    for (int i = 0; i < n; ++i) {
        const float v = imgs[(size_t)i * (size_t)w * (size_t)h];
        if (v < 0) {
            c->status[(size_t)i] = SIFT_HIP_EPRECONDITION;
            if (rc == SIFT_HIP_OK && err && errlen > 0) std::snprintf(err, (size_t)errlen, "fake precondition, frame %d", i);
            rc = SIFT_HIP_EPRECONDITION;
            continue;
        }
        const size_t before = c->kp.size();
        // (a frame whose SECOND pixel is >= 1000 asks for that many records: the large results sift::Sift::collect() shares out)
        const float second = (size_t)w * (size_t)h > 1 ? imgs[(size_t)i * (size_t)w * (size_t)h + 1] : 0.0f;
        fake_points(v, c->kp, c->desc, second >= 1000.0f ? (int)second : 0);
        c->counts[(size_t)i] = (int32_t)(c->kp.size() - before);
    }
    c->have = true;
    return rc;
}
int sift_hip_result_images(sift_hip_ctx* c) { return c->have ? (int)c->status.size() : -1; }
int sift_hip_result_status(sift_hip_ctx* c, int32_t* st, int cap) { if (cap < (int)c->status.size()) return SIFT_HIP_EINVAL; std::copy(c->status.begin(), c->status.end(), st); return SIFT_HIP_OK; }
int sift_hip_result_counts(sift_hip_ctx* c, int32_t* ct, int cap) { if (cap < (int)c->counts.size()) return SIFT_HIP_EINVAL; std::copy(c->counts.begin(), c->counts.end(), ct); return SIFT_HIP_OK; }
int64_t sift_hip_result_total(sift_hip_ctx* c) { return (int64_t)c->kp.size(); }
int sift_hip_result_device(sift_hip_ctx* c, const void** kp, const void** desc) { *kp = c->kp.data(); *desc = c->desc.data(); return SIFT_HIP_OK; }
static bool carried(int p, float f) { uint32_t b; std::memcpy(&b, &f, 4); return (p & 7) != 7 && b != 0u; }
int sift_hip_result_sparse_size(sift_hip_ctx* c, int64_t* n_values, int* lossless) {
    int64_t n = 0;
    for (size_t i = 0; i < c->desc.size(); ++i) n += carried((int)(i & 127), c->desc[i]);
    c->nnz = n; *n_values = n;
    if (lossless) *lossless = 1;
    return SIFT_HIP_OK;
}
int sift_hip_result_sparse_pack(sift_hip_ctx* c, void* rec, void* val) {
    uint8_t* r = static_cast<uint8_t*>(rec);
    float* v = static_cast<float*>(val);
    for (size_t k = 0; k < c->kp.size(); ++k, r += 34) {
        std::memcpy(r, &c->kp[k], 20);
        std::memset(r + 20, 0, 14);
        for (int p = 0; p < 128; ++p)
            if (carried(p, c->desc[k * 128 + (size_t)p])) {
                const int j = (p >> 3) * 7 + (p & 7);
                r[20 + (j >> 3)] |= (uint8_t)(1u << (j & 7));
                *v++ = c->desc[k * 128 + (size_t)p];
            }
    }
    return SIFT_HIP_OK;
}
int sift_hip_sparse_unpack(sift_hip_ctx*, const void* rec, const void* val, int64_t n, void* kp, void* desc) {
    const uint8_t* r = static_cast<const uint8_t*>(rec);
    const float* v = static_cast<const float*>(val);
    float* d = static_cast<float*>(desc);
    for (int64_t k = 0; k < n; ++k, r += 34, d += 128) {
        std::memcpy(static_cast<char*>(kp) + k * 20, r, 20);
        for (int p = 0; p < 128; ++p) {
            const int j = (p >> 3) * 7 + (p & 7);
            d[p] = ((p & 7) != 7 && (r[20 + (j >> 3)] >> (j & 7) & 1)) ? *v++ : 0.0f;
        }
    }
    return SIFT_HIP_OK;
}
int sift_hip_internal_copy(void*, const void* src, void* dst, size_t bytes) { std::memcpy(dst, src, bytes); return 0; }
// ---- what include/sift/sift.hpp calls beside the above (collect_main.cpp) ----
int sift_hip_calculate_batch_u8(sift_hip_ctx* c, const uint8_t* imgs, int n, int w, int h, const sift_hip_params* p, char* err, int errlen) {
    std::vector<float> f((size_t)n * (size_t)w * (size_t)h);
    for (size_t i = 0; i < f.size(); ++i) f[i] = (float)imgs[i];
    return sift_hip_calculate_batch(c, f.data(), n, w, h, p, err, errlen);
}
int sift_hip_result_copy(sift_hip_ctx* c, sift_hip_keypoint* kp, float* desc) {
    if (kp) std::copy(c->kp.begin(), c->kp.end(), kp);
    if (desc) std::copy(c->desc.begin(), c->desc.end(), desc);
    return SIFT_HIP_OK;
}
int sift_hip_result_copy_sparse(sift_hip_ctx* c, void* rec, float* val) { return sift_hip_result_sparse_pack(c, rec, val); }
int sift_hip_sparse_unpack_host(const void* rec, const float* val, int64_t n, sift_hip_keypoint* kp, float* desc, int) {
    // (the caller may pass either output; the device form above wants both)
    std::vector<sift_hip_keypoint> k2(kp ? 0 : (size_t)n);
    std::vector<float> d2(desc ? 0 : (size_t)n * 128);
    return sift_hip_sparse_unpack(nullptr, rec, val, n, kp ? (void*)kp : (void*)k2.data(), desc ? (void*)desc : (void*)d2.data());
}
int sift_hip_image_dims(sift_hip_ctx*, int* w, int* h) { *w = *h = 0; return SIFT_HIP_EINVAL; }
int sift_hip_image_copy(sift_hip_ctx*, int, float*) { return SIFT_HIP_EINVAL; }
struct sift_hip_gate { int device; };
int sift_hip_gate_create(int device, sift_hip_gate** out) { *out = new sift_hip_gate{device}; return SIFT_HIP_OK; }
void sift_hip_gate_destroy(sift_hip_gate* g) { delete g; }
int sift_hip_set_gate(sift_hip_ctx*, sift_hip_gate*) { return SIFT_HIP_OK; }
void* sift_hip_host_alloc(size_t n) { return std::malloc(n ? n : 1); }
void sift_hip_host_free(void* p) { std::free(p); }
}
