// Driver of the sanitizer build of include/sift/sift.hpp's collect() - the persistent worker pool that expands large results
// (condition variable, job id, parked workers: ADVICE r05) - and of sift_amd/csrc/launch_guard.cpp as it ships (the per-device
// launch locks, the table of cached function objects, the per-thread launch error), on the fake runtime (fake_hip.cpp compiled
// with -DSIFT_FAKE_WITH_REAL_LAUNCH_GUARD).  tests/test_host_tsan.py builds it with -fsanitize=thread and =address,undefined.
//   * results of 3 ... 20 000 points through one Sift object, in an order that starts the pool late, grows it never, leaves it
//     parked between calls and parks it at destruction - every point and every descriptor float compared with what the fake
//     context layer defines;
//   * two Sift objects on two threads at once (each has a pool of its own);
//   * two threads on two devices resolving and "launching" the same kernels through cached_function under their devices' locks,
//     a third thread on the first device's lock: every (device, kernel) resolves to one object, errors stay with their thread.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/sift/sift.hpp"
#include "../../sift_amd/csrc/launch_guard.h"

namespace sift_hip {
hipFunction_t cached_function(const void* host_stub);
hipError_t take_launch_error();
void note_launch_error(hipError_t e);
}

namespace {
std::atomic<int> fails{0};
#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed %s:%d: ", __FILE__, __LINE__); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); ++fails; } } while (0)

constexpr int W = 8, H = 4;

// what the fake context layer returns for a frame with first pixel v and `count` requested records (fake_hip.cpp: fake_points)
void check_points(const std::vector<sift::InterestPoint>& pts, float v, int count, const char* what) {
    const int n = count > 0 ? count : 1 + ((int)v % 7 + 7) % 7;
    CHECK((int)pts.size() == n, "%s: %zu points, expected %d", what, pts.size(), n);
    for (int j = 0; j < n && j < (int)pts.size(); ++j) {
        const sift::InterestPoint& p = pts[(size_t)j];
        bool ok = p.scale == v && p.orientation == 177.5f && p.loc.x == (u16_t)((int)v + j) && p.loc.y == (u16_t)j && p.descriptors.size() == 128;
        if (ok) {
            float want[128] = {0};
            for (int k = 0; k < 20; ++k) {
                const int q = (j + 3 * k) % 128;
                if ((q & 7) != 7) want[q] = v + 0.25f * (float)k;
            }
            ok = std::memcmp(want, p.descriptors.data(), sizeof(want)) == 0;
        }
        if (!ok) { CHECK(false, "%s: point %d of %d differs", what, j, n); break; }
    }
}

void run_object(int device, unsigned seed, int rounds) {
    sift::Sift s(3, 3, 1.6f, 1.41421356f, false, device);
    const int counts[] = {0, 0, 6000, 0, 20000, 4096, 4095, 0, 20000, 1000, 12345};   // (0: the frame's own 1 .. 7 records)
    for (int r = 0; r < rounds; ++r) {
        const int count = counts[(r + (int)seed) % (int)(sizeof(counts) / sizeof(counts[0]))];
        sift::Image2f img(W, H);
        for (int i = 0; i < W * H; ++i) img.data()[i] = 1.0f;
        const float v = (float)(1 + (r * 7 + (int)seed) % 40);
        img.data()[0] = v;
        img.data()[1] = count > 0 ? (float)count : 1.0f;
        char what[64];
        std::snprintf(what, sizeof(what), "device %d round %d count %d", device, r, count);
        check_points(s.calculate(img), v, count, what);
    }
}   // the object goes with its workers parked

__attribute__((noinline)) void stub_a() {}
__attribute__((noinline)) void stub_b() {}
__attribute__((noinline)) void stub_c() {}

void run_launcher(int device, int iterations, hipFunction_t (&seen)[3]) {
    (void)sift_hip::set_device_tracked(device);
    const void* stubs[3] = {reinterpret_cast<const void*>(&stub_a), reinterpret_cast<const void*>(&stub_b), reinterpret_cast<const void*>(&stub_c)};
    for (int it = 0; it < iterations; ++it) {
        const int k = it % 3;
        hipFunction_t f;
        {
            sift_hip::LaunchGuard guard;   // the launch path: the lookup runs under the device's lock
            f = sift_hip::cached_function(stubs[k]);
        }
        CHECK(f != nullptr, "device %d: no function object", device);
        if (seen[k] == nullptr) seen[k] = f;
        CHECK(seen[k] == f, "device %d kernel %d: the cached object changed", device, k);
        if (it % 97 == 0) {   // an error noted on this thread stays on this thread, first one wins, taking clears it
            sift_hip::note_launch_error(device == 0 ? hipErrorInvalidConfiguration : hipErrorInvalidDeviceFunction);
            sift_hip::note_launch_error(hipErrorUnknown);
            const hipError_t e = sift_hip::take_launch_error();
            CHECK(e == (device == 0 ? hipErrorInvalidConfiguration : hipErrorInvalidDeviceFunction), "device %d: error %d", device, (int)e);
            CHECK(sift_hip::take_launch_error() == hipSuccess, "device %d: the error was not cleared", device);
        }
    }
}
}  // namespace

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 22;
    run_object(0, 0, rounds);                                   // one object, pool started by the third call
    {
        std::thread a(run_object, 0, 3u, rounds), b(run_object, 1, 5u, rounds);   // two objects, two threads, a pool each
        a.join();
        b.join();
    }
    {
        hipFunction_t s0[3] = {nullptr, nullptr, nullptr}, s1[3] = {nullptr, nullptr, nullptr}, s2[3] = {nullptr, nullptr, nullptr};
        std::thread a([&] { run_launcher(0, 20000, s0); }), b([&] { run_launcher(1, 20000, s1); }), c([&] { run_launcher(0, 20000, s2); });
        a.join();
        b.join();
        c.join();
        for (int k = 0; k < 3; ++k) {
            CHECK(s0[k] == s2[k], "kernel %d: two threads of device 0 hold different objects", k);
            CHECK(s0[k] != s1[k], "kernel %d: devices 0 and 1 share an object", k);
        }
    }
    if (fails.load()) {
        std::fprintf(stderr, "%d checks failed\n", fails.load());
        return 1;
    }
    std::printf("collect ok\n");
    return 0;
}
