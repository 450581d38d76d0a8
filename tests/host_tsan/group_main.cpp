// Driver of the sanitizer build of sift_hip_group (tests/test_host_tsan.py): batches of random sizes through groups of 1 - 4
// shards on the fake runtime (fake_hip.cpp), calculate / submit-submit-collect-collect with a larger second batch / failing
// frames / empty shards, every gathered result compared with the lists the fake context layer defines for those frames.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/sift_hip.h"

extern "C" long long fake_live_allocs(void);

namespace {
constexpr int W = 8, H = 4;

struct Expect {
    std::vector<sift_hip_keypoint> kp;
    std::vector<float> desc;
    std::vector<int32_t> counts, status;
    int rc = SIFT_HIP_OK;
};

// what the fake context layer returns for a frame whose first pixel is v (fake_hip.cpp: fake_points)
Expect expect_for(const std::vector<float>& frames, int n) {
    Expect e;
    for (int i = 0; i < n; ++i) {
        const float v = frames[(size_t)i * W * H];
        if (v < 0) { e.status.push_back(SIFT_HIP_EPRECONDITION); e.counts.push_back(0); if (e.rc == SIFT_HIP_OK) e.rc = SIFT_HIP_EPRECONDITION; continue; }
        const int cnt = 1 + ((int)v % 7 + 7) % 7;
        e.status.push_back(0);
        e.counts.push_back(cnt);
        for (int j = 0; j < cnt; ++j) {
            sift_hip_keypoint r;
            std::memset(&r, 0, sizeof(r));
            r.scale = v; r.orientation = 177.5f; r.x = (uint16_t)((int)v + j); r.y = (uint16_t)j; r.octave = 1; r.index = 1; r.has_descriptor = 1;
            e.kp.push_back(r);
            const size_t base = e.desc.size();
            e.desc.resize(base + 128, 0.0f);
            for (int k = 0; k < 20; ++k) {
                const int p = (j + 3 * k) % 128;
                if ((p & 7) != 7) e.desc[base + (size_t)p] = v + 0.25f * (float)k;
            }
        }
    }
    return e;
}

int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed %s:%d: ", __FILE__, __LINE__); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); ++fails; } } while (0)

void check_result(sift_hip_group* g, const Expect& e, int n, int rc, const char* what) {
    CHECK(rc == e.rc, "%s: rc %d, expected %d", what, rc, e.rc);
    CHECK(sift_hip_group_result_images(g) == n, "%s: images %d vs %d", what, sift_hip_group_result_images(g), n);
    std::vector<int32_t> st((size_t)n), ct((size_t)n);
    CHECK(sift_hip_group_result_status(g, st.data(), n) == SIFT_HIP_OK && st == e.status, "%s: status", what);
    CHECK(sift_hip_group_result_counts(g, ct.data(), n) == SIFT_HIP_OK && ct == e.counts, "%s: counts", what);
    const long long total = sift_hip_group_result_total(g);
    CHECK(total == (long long)e.kp.size(), "%s: total %lld vs %zu", what, total, e.kp.size());
    if (total != (long long)e.kp.size() || total == 0) return;
    std::vector<sift_hip_keypoint> kp((size_t)total);
    std::vector<float> desc((size_t)total * 128);
    CHECK(sift_hip_group_result_copy(g, kp.data(), desc.data()) == SIFT_HIP_OK, "%s: result_copy", what);
    CHECK(std::memcmp(kp.data(), e.kp.data(), kp.size() * sizeof(sift_hip_keypoint)) == 0, "%s: keypoint records differ", what);
    CHECK(std::memcmp(desc.data(), e.desc.data(), desc.size() * sizeof(float)) == 0, "%s: descriptors differ", what);
}
}  // namespace

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 40;
    std::mt19937 rng(12345);
    sift_hip_params prm{};
    prm.dogs_per_epoch = 3; prm.octaves = 3; prm.sigma = 1.6f; prm.k = 1.41421354f;
    char err[512];
    for (int shards = 1; shards <= 4; ++shards)
        for (int wire = 0; wire <= 2; ++wire) {
            std::vector<int> devices((size_t)shards, 0);
            if (shards >= 3) devices[(size_t)shards - 1] = 1;   // one shard on another (fake) device: the peer path
            sift_hip_group* g = nullptr;
            if (sift_hip_group_create(devices.data(), shards, &g, err, sizeof(err)) != SIFT_HIP_OK) { std::fprintf(stderr, "create: %s\n", err); return 1; }
            sift_hip_group_set_option(g, "gather_transport", 0);   // copies (no RCCL on the CPU)
            sift_hip_group_set_option(g, "gather_wire", wire);
            auto make = [&](int n, bool with_failure) {
                std::vector<float> f((size_t)n * W * H, 1.0f);
                for (int i = 0; i < n; ++i) f[(size_t)i * W * H] = (float)(rng() % 200);
                if (with_failure && n > 1) f[(size_t)(rng() % (unsigned)n) * W * H] = -1.0f;
                return f;
            };
            for (int r = 0; r < rounds; ++r) {
                // one batch at a time
                const int n = 1 + (int)(rng() % 9);
                const std::vector<float> f = make(n, r % 7 == 3);
                int rc = sift_hip_group_calculate(g, f.data(), n, W, H, &prm, err, sizeof(err));
                check_result(g, expect_for(f, n), n, rc, "calculate");
                // two in flight, the second one larger (its buffers grow while the first one's lists are under way)
                const int n1 = 1 + (int)(rng() % 4), n2 = n1 + 3 + (int)(rng() % 12);
                const std::vector<float> f1 = make(n1, false), f2 = make(n2, r % 5 == 2);
                CHECK(sift_hip_group_submit(g, f1.data(), n1, W, H, &prm, err, sizeof(err)) == SIFT_HIP_OK, "submit 1: %s", err);
                CHECK(sift_hip_group_submit(g, f2.data(), n2, W, H, &prm, err, sizeof(err)) == SIFT_HIP_OK, "submit 2: %s", err);
                CHECK(sift_hip_group_submit(g, f2.data(), n2, W, H, &prm, err, sizeof(err)) == SIFT_HIP_EINVAL, "a third batch in flight must be refused");
                rc = sift_hip_group_collect(g, err, sizeof(err));
                check_result(g, expect_for(f1, n1), n1, rc, "collect 1");
                rc = sift_hip_group_collect(g, err, sizeof(err));
                check_result(g, expect_for(f2, n2), n2, rc, "collect 2");
                CHECK(sift_hip_group_collect(g, err, sizeof(err)) == SIFT_HIP_EINVAL, "nothing left to collect");
            }
            // destroyed with a batch still in flight: it runs to its end first
            const std::vector<float> f = make(5, false);
            CHECK(sift_hip_group_submit(g, f.data(), 5, W, H, &prm, err, sizeof(err)) == SIFT_HIP_OK, "last submit");
            sift_hip_group_destroy(g);
        }
    CHECK(fake_live_allocs() == 0, "%lld device / pinned allocations were never freed", fake_live_allocs());
    if (fails) { std::fprintf(stderr, "%d checks failed\n", fails); return 2; }
    std::printf("group ok\n");
    return 0;
}
