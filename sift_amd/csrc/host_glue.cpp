// See host_glue.h.  Compiled with -ffp-contract=off: the tap arithmetic must round exactly like the
// reference's scalar SSE code.
#include "host_glue.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>

namespace sift_hip {

bool gauss_taps(float sigma_f, std::vector<float>& taps, int& radius) {
    const double std_dev = (double)sigma_f;
    taps.clear();
    if (!(std_dev >= 0.0)) return false;
    if (std_dev > 0.0) {
        const float sigma = (float)std_dev;
        const float sigma2 = (float)(-0.5 / sigma / sigma);
        const float norm = (float)(1.0 / std::sqrt(2.0 * M_PI) / sigma);
        radius = (int)(3.0 * std_dev + 0.5);
        if (radius == 0) radius = 1;
        taps.reserve((size_t)radius * 2 + 1);
        for (float x = -(float)radius; x <= (float)radius; ++x) {
            const float x2 = x * x;
            taps.push_back(norm * expf(x2 * sigma2));
        }
    } else {
        radius = 0;
        taps.push_back(1.0f);
    }
    float sum = 0.0f;
    for (float v : taps) sum += v;
    const float scale = 1.0f / sum;
    for (float& v : taps) v = v * scale;
    return true;
}

std::vector<int> resize_index_map(int wold, int wnew) {
    std::vector<int> idx((size_t)wnew);
    if (wnew == 1) {
        idx[0] = 0;
        return idx;
    }
    const double dx = (double)(wold - 1) / (double)(wnew - 1);
    double x = 0.5;
    for (int i = 0; i < wnew; ++i, x += dx) idx[(size_t)i] = (int)x;
    return idx;
}

namespace {
struct Proxy {
    uint32_t idx;
    uint32_t filtered;
};
inline bool cmp_by_filter(const Proxy& a, const Proxy& b) { return !a.filtered && b.filtered; }
}  // namespace

void sort_by_filter(const uint8_t* flags, int n, std::vector<uint32_t>& perm) {
    std::vector<Proxy> v((size_t)n);
    for (int i = 0; i < n; ++i) v[(size_t)i] = Proxy{(uint32_t)i, flags[i] ? 1u : 0u};
    // The sequence of comparisons and moves std::sort performs depends only on the comparator's
    // answers, so sorting proxies yields the permutation the reference's vector undergoes.
    std::sort(v.begin(), v.end(), cmp_by_filter);
    perm.resize((size_t)n);
    for (int i = 0; i < n; ++i) perm[(size_t)i] = v[(size_t)i].idx;
}

void cleanup_survivors(const uint8_t* flags, int n, std::vector<uint32_t>& survivors) {
    std::vector<Proxy> v((size_t)n);
    for (int i = 0; i < n; ++i) v[(size_t)i] = Proxy{(uint32_t)i, flags[i] ? 1u : 0u};
    std::sort(v.begin(), v.end(), cmp_by_filter);
    auto it = std::find_if(v.begin(), v.end(), [](const Proxy& p) { return p.filtered != 0; });
    const uint16_t size = (uint16_t)std::distance(v.begin(), it);  // u16_t size (sift.cpp:41)
    survivors.resize(size);
    for (size_t i = 0; i < size; ++i) survivors[i] = v[i].idx;
}

void parallel_for(int n, int threads, void (*fn)(int, void*), void* arg) {
    if (threads > n) threads = n;
    if (threads <= 1) {
        for (int i = 0; i < n; ++i) fn(i, arg);
        return;
    }
    std::atomic<int> next(0);
    std::vector<std::thread> pool;
    pool.reserve((size_t)threads);
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&]() {
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= n) break;
                fn(i, arg);
            }
        });
    for (auto& th : pool) th.join();
}

}  // namespace sift_hip
