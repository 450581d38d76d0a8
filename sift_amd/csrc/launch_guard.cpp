// The per-device launch locks every HIP call of the library is made under (launch_guard.h), the calling thread's tracked device and
// the table of function objects the kernels are launched through (launch_cache.h).  Plain host code in a file of its own (round 6;
// until then part of kernels_io.hip) so that tests/test_host_tsan.py can compile it as it ships against the fake runtime and run
// threads of two devices through it under ThreadSanitizer / AddressSanitizer.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <unordered_map>

#include "launch_guard.h"

namespace sift_hip {

hipFunction_t cached_function(const void* host_stub);
hipError_t take_launch_error();
void note_launch_error(hipError_t e);
hipError_t combined_last_error();

// ---- launch locks (launch_guard.h) -------------------------------------------------------------------------------------
constexpr int kMaxLockDevices = 64;
std::recursive_mutex& launch_lock_of(int device) {
    static std::recursive_mutex m[kMaxLockDevices];
    return m[(unsigned)device % (unsigned)kMaxLockDevices];
}
static thread_local int t_device = 0;
int set_device_tracked(int device) {
    const hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) t_device = device;
    return (int)e;
}
int tracked_device() { return t_device; }
int current_device_refreshed() {
    int d = t_device;
    if (hipGetDevice(&d) == hipSuccess) t_device = d;
    else (void)hipGetLastError();
    return t_device;
}
std::recursive_mutex& launch_lock() { return launch_lock_of(t_device); }
static std::atomic<long long> g_lock_wait_ns{0};
double launch_lock_wait_ms() { return (double)g_lock_wait_ns.load(std::memory_order_relaxed) / 1e6; }
static void lock_accounted(std::recursive_mutex& m) {
    if (m.try_lock()) return;
    const auto t0 = std::chrono::steady_clock::now();
    m.lock();
    g_lock_wait_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
}
LaunchGuard::LaunchGuard() : m(launch_lock()) { lock_accounted(m); }

// ---- function objects of the library's kernels, per device (launch_cache.h) ------------------------------------------------
static std::unordered_map<const void*, hipFunction_t> g_functions[kMaxLockDevices];   // each under its device's launch lock
hipFunction_t cached_function(const void* host_stub) {
    auto& table = g_functions[(unsigned)t_device % (unsigned)kMaxLockDevices];
    const auto it = table.find(host_stub);
    if (it != table.end()) return it->second;
    hipFunction_t f = nullptr;
    if (hipGetFuncBySymbol(&f, host_stub) != hipSuccess || !f) {
        (void)hipGetLastError();
        return nullptr;
    }
    table.emplace(host_stub, f);
    return f;
}
static thread_local hipError_t t_launch_error = hipSuccess;
void note_launch_error(hipError_t e) { if (t_launch_error == hipSuccess) t_launch_error = e; }
hipError_t take_launch_error() {
    const hipError_t e = t_launch_error;
    t_launch_error = hipSuccess;
    return e;
}
hipError_t combined_last_error() {
    const hipError_t mine = take_launch_error(), runtime = hipGetLastError();
    return mine != hipSuccess ? mine : runtime;
}
LaunchGuard::LaunchGuard(int device) : m(launch_lock_of(device)) { lock_accounted(m); }

}  // namespace sift_hip
