// Kernel launches without the runtime's per-launch lookup (round 5).
//
// hipLaunchKernel(host stub, ...) - what `kernel<<<...>>>` and hipLaunchKernelGGL compile to - looks the stub's address up in
// the runtime's table of statically registered functions on EVERY launch, for the current device.  That lookup is where
// multi-threaded hosts of this library died on HIP 7.2 (SEGV a few frames below hipLaunchKernel: the lookup returned a null
// function object while another host thread was inside the runtime; common.h, tools/example_loop.sh, profiles/r04_soak.txt).
// Here the lookup is made ONCE per (device, kernel) - hipGetFuncBySymbol, under the device's launch lock - and its result,
// a hipFunction_t, is kept; every launch then goes through hipExtModuleLaunchKernel, which takes the function object itself.
// The kernel's parameters are converted to the kernel's own parameter types (what the <<<>>> call would do) and handed over
// as the array of pointers the module launch wants.
//
// -DSIFT_HIP_STATIC_LAUNCH restores the runtime's own path (A/B builds: tools/example_loop.sh).
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <tuple>
#include <utility>

#include "launch_guard.h"

namespace sift_hip {

// (device of the calling thread, host stub) -> function object; resolved on first use.  Call with the device's launch lock held.
hipFunction_t cached_function(const void* host_stub);
// The last error of the launch path on this thread (hipSuccess if none), cleared by the call: the module launch returns its
// error instead of leaving it for hipGetLastError, and the library's callers check once per stage.
hipError_t take_launch_error();
void note_launch_error(hipError_t e);

template <class... KArgs, class... Args, size_t... I>
inline void launch_cached_impl(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t s, hipEvent_t ev_start,
                               hipEvent_t ev_stop, std::index_sequence<I...>, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel launched with the wrong number of arguments");
    std::tuple<KArgs...> a{static_cast<KArgs>(std::forward<Args>(args))...};
    void* ptrs[sizeof...(KArgs) > 0 ? sizeof...(KArgs) : 1] = {const_cast<void*>(static_cast<const void*>(&std::get<I>(a)))...};
    // the module launch takes the grid in THREADS per dimension, as 32-bit numbers: a launch of 2^32 threads or more along a
    // dimension (the runtime's own path accepted 2^31 blocks) must fail, not wrap
    if ((unsigned long long)grid.x * block.x > 0xffffffffull || (unsigned long long)grid.y * block.y > 0xffffffffull ||
        (unsigned long long)grid.z * block.z > 0xffffffffull) {
        note_launch_error(hipErrorInvalidConfiguration);
        return;
    }
    LaunchGuard guard;
    hipFunction_t f = cached_function(reinterpret_cast<const void*>(kernel));
    if (!f) { note_launch_error(hipErrorInvalidDeviceFunction); return; }
    const hipError_t e = hipExtModuleLaunchKernel(f, grid.x * block.x, grid.y * block.y, grid.z * block.z, block.x, block.y, block.z, shmem, s,
                                                  ptrs, nullptr, ev_start, ev_stop, 0);
    if (e != hipSuccess) note_launch_error(e);
}

template <class... KArgs, class... Args>
inline void launch_cached(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t s, hipEvent_t ev_start,
                          hipEvent_t ev_stop, Args&&... args) {
    launch_cached_impl(kernel, grid, block, shmem, s, ev_start, ev_stop, std::index_sequence_for<KArgs...>{}, std::forward<Args>(args)...);
}

}  // namespace sift_hip
