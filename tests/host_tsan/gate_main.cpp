// Driver of the sanitizer build of sift_amd/csrc/phase_gate.h (tests/test_host_tsan.py): two, three and four "contexts" (host
// threads) take tickets from one gate and go through a batch's calls - begin_batch, mark P / E, before_cleanup,
// before_descriptors, mark D, finish - in both schedules, some batches ending early (finish alone), on the fake runtime whose
// events are always complete.  What is checked: no data race, no deadlock (the test has a timeout), every ticket handed out once.
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../sift_amd/csrc/launch_guard.h"
namespace sift_hip { using ApiGuard = LaunchGuard; }
#include "../../sift_amd/csrc/phase_gate.h"

int main() {
    for (int schedule = 0; schedule <= 1; ++schedule)
        for (int contexts = 2; contexts <= sift_hip::PhaseGate::kMaxContexts; ++contexts) {
            sift_hip::PhaseGate gate;
            gate.set_schedule(schedule);
            const int batches = 300;
            std::vector<std::atomic<int>> seen((size_t)batches * contexts);
            for (auto& s : seen) s = 0;
            std::vector<std::thread> th;
            for (int c = 0; c < contexts; ++c)
                th.emplace_back([&, c] {
                    hipStream_t s = nullptr;
                    for (int b = 0; b < batches; ++b) {
                        const long long g = gate.begin_batch(s);
                        if (g >= 0 && g < (long long)seen.size()) seen[(size_t)g]++;
                        if ((b + c) % 11 == 5) { gate.finish(g, s); continue; }   // a batch that fails half way
                        gate.mark(g, sift_hip::PhaseGate::kP, s);
                        gate.mark(g, sift_hip::PhaseGate::kE, s);
                        gate.before_cleanup(g, s);
                        gate.before_descriptors(g, s);
                        gate.mark(g, sift_hip::PhaseGate::kD, s);
                        gate.finish(g, s);
                    }
                });
            for (auto& t : th) t.join();
            for (size_t i = 0; i < seen.size(); ++i)
                if (seen[i] != 1) { std::fprintf(stderr, "schedule %d, %d contexts: ticket %zu handed out %d times\n", schedule, contexts, i, seen[i].load()); return 2; }
        }
    std::printf("gate ok\n");
    return 0;
}
