// Host threads and the HIP runtime: what a launch must not run beside (see common.h for the history).  One lock PER DEVICE
// since round 4: the crashes were a launch of one thread beside another thread's runtime copy, allocation or stream / event
// call ON THE SAME DEVICE; rounds 2 - 3 used one process-wide lock, which made every launch of an 8-GPU group of shard
// threads queue behind the other seven GPUs' launches - and made a deadlock possible: a shard that grows a buffer while its
// RCCL send of the previous batch is still unmatched sits in hipFree (which waits for the whole device, i.e. for that send)
// HOLDING the lock every other shard needs to finish its batch and report, which is what the receive waits for.  With one
// lock per device the other shards (other devices) and the gather thread get on, the receive is posted and hipFree returns.
// The time threads spend waiting for a lock is accounted (launch_lock_wait_ms, sift_hip_lock_wait_ms).
//
// Measured in round 4 and NOT kept (tools/example_loop.sh, 100 - 150 runs of examples/sift_multi_gpu.cpp each, one GPU):
//   * growing buffers retiring their old allocation instead of hipFree (freed when the host is idle): 13 - 22 % of the runs died
//     below hipLaunchKernel against 0 - 4 % - the device-wide wait inside hipFree, under the lock, keeps the threads apart
//     while a context's buffers settle, which is when the runtime is most fragile;
//   * every wait of the library as an event polled under the lock (no hipStreamSynchronize / hipEventSynchronize beside another
//     thread's launch): no change (19 of 150);
//   * asking the runtime for the current device in front of every launch (hipGetDevice): no change either; the device is kept
//     in a thread-local all the same.
#pragma once
#include <mutex>

namespace sift_hip {

// The calling thread's device as THIS LIBRARY set it last (every entry point sets its context's device; a thread that never
// did is on device 0, the runtime's default).
int set_device_tracked(int device);           // hipSetDevice + the thread-local; returns the hipError_t as an int
int tracked_device();
// ... and as the RUNTIME reports it (one hipGetDevice; refreshes the thread-local): for entry points that take no context and
// may be called by a thread whose device the host set itself - torch.cuda.set_device, a gather thread
int current_device_refreshed();

std::recursive_mutex& launch_lock_of(int device);
std::recursive_mutex& launch_lock();          // of the calling thread's tracked device
double launch_lock_wait_ms();                 // total over all threads since the process started

struct LaunchGuard {                          // lock of the current device (or of `device`), waiting time accounted
    std::recursive_mutex& m;
    LaunchGuard();
    explicit LaunchGuard(int device);
    ~LaunchGuard() { m.unlock(); }
    LaunchGuard(const LaunchGuard&) = delete;
    LaunchGuard& operator=(const LaunchGuard&) = delete;
};

}  // namespace sift_hip
