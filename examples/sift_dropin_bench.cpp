// The drop-in boundary as the reference defines it, timed: sift::Sift::calculate(Image2f&) - a float host image in, a
// std::vector<InterestPoint> with one heap std::vector<f32_t> of 128 floats per point out (/root/reference/sift.hpp:78,
// interestpoint.hpp:46, main.cpp:52-57) - called in a loop on one frame, then by two gated Sift objects on two threads.
// Prints ONE JSON line; bench.py runs it outside its timed region and reports the figures beside `single_frame_ms`.
//   g++ -O2 -std=c++17 -pthread -Iinclude examples/sift_dropin_bench.cpp -Lsift_amd/lib -lsift_hip -Wl,-rpath,$PWD/sift_amd/lib -o sift_dropin_bench
//   ./sift_dropin_bench frame.f32 1920 1080 [iterations=50] [octaves=4] [dogs=3]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <vector>

#include "sift/sift.hpp"

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);   // before the first HIP call: two contexts side by side want hardware queues of their own
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s frame.f32 width height [iterations] [octaves] [dogs]\n", argv[0]);
        return 1;
    }
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    const int iters = argc > 4 ? std::atoi(argv[4]) : 50;
    const int octaves = argc > 5 ? std::atoi(argv[5]) : 4, dogs = argc > 6 ? std::atoi(argv[6]) : 3;
    try {
        sift::Image2f frame(w, h);
        {
            std::ifstream f(argv[1], std::ios::binary);
            f.read(reinterpret_cast<char*>(frame.data()), (std::streamsize)((size_t)w * (size_t)h * sizeof(float)));
            if (!f) throw std::runtime_error("cannot read the frame");
        }
        // 1. one object, one thread: the reference's own call pattern (main.cpp:56-57).  A call's time runs from the call to
        //    the moment the caller has dropped the result again (freeing ~20 k heap blocks is part of what this boundary costs).
        std::vector<double> call_ms, drop_ms;
        size_t points = 0, with_desc = 0;
        {
            sift::Sift sift((u16_t)dogs, (u16_t)octaves);
            for (int i = 0; i < 3; ++i) { sift::Image2f img = frame; (void)sift.calculate(img); }   // warm-up: plan, buffers, first launches
            for (int i = 0; i < iters; ++i) {
                const double t0 = now_ms();
                double t1;
                {
                    std::vector<sift::InterestPoint> pts = sift.calculate(frame);
                    t1 = now_ms();
                    points = pts.size();
                    with_desc = 0;
                    for (const auto& p : pts) with_desc += p.descriptors.size() == 128;
                }
                const double t2 = now_ms();
                call_ms.push_back(t1 - t0);
                drop_ms.push_back(t2 - t1);
            }
        }
        // 1b. where a call's time goes: the same frame through the C ABI piece by piece (what Sift::calculate does inside)
        double abi_calc_ms = 0, abi_fetch_ms = 0;
        {
            sift_hip_ctx* ctx = nullptr;
            char err[256] = "";
            if (sift_hip_create(0, &ctx, err, sizeof(err)) != SIFT_HIP_OK) throw std::runtime_error(err);
            sift_hip_params p{};
            p.dogs_per_epoch = (u16_t)dogs; p.octaves = (u16_t)octaves; p.sigma = 1.6f; p.k = std::sqrt(2.0f); p.subpixel = 0;
            std::vector<unsigned char> rec;
            std::vector<float> val;
            std::vector<double> tc, tf;
            for (int i = 0; i < 3 + iters / 2; ++i) {
                const double t0 = now_ms();
                if (sift_hip_calculate_batch(ctx, frame.data(), 1, w, h, &p, err, sizeof(err)) != SIFT_HIP_OK) throw std::runtime_error(err);
                const double t1 = now_ms();
                int64_t nnz = 0;
                int lossless = 0;
                const long long n = sift_hip_result_total(ctx);
                sift_hip_result_sparse_size(ctx, &nnz, &lossless);
                if (rec.size() < (size_t)n * 34) rec.resize((size_t)n * 34);
                if (val.size() < (size_t)nnz + 8) val.resize((size_t)nnz + 8);
                sift_hip_result_copy_sparse(ctx, rec.data(), val.data());
                const double t2 = now_ms();
                if (i >= 3) { tc.push_back(t1 - t0); tf.push_back(t2 - t1); }
            }
            abi_calc_ms = median(tc);
            abi_fetch_ms = median(tf);
            sift_hip_destroy(ctx);
        }
        // 2. two objects joined by a gate, one host thread each, frames handed out alternately
        double pair_ms = 0;
        long long pair_points = 0;
        {
            sift_hip_gate* gate = nullptr;
            if (sift_hip_gate_create(0, &gate) != SIFT_HIP_OK) throw std::runtime_error("sift_hip_gate_create failed");
            {
                sift::Sift a((u16_t)dogs, (u16_t)octaves), b((u16_t)dogs, (u16_t)octaves);
                a.join(gate);
                b.join(gate);
                for (sift::Sift* s : {&a, &b})
                    for (int i = 0; i < 3; ++i) { sift::Image2f img = frame; (void)s->calculate(img); }
                std::atomic<int> next{0};
                std::atomic<long long> total{0};
                auto worker = [&](sift::Sift& s) {
                    sift::Image2f img = frame;
                    for (int k = next++; k < 2 * iters; k = next++) total += (long long)s.calculate(img).size();
                };
                const double t0 = now_ms();
                std::thread ta(worker, std::ref(a)), tb(worker, std::ref(b));
                ta.join();
                tb.join();
                pair_ms = now_ms() - t0;
                pair_points = total.load();
                a.join(nullptr);
                b.join(nullptr);
            }
            sift_hip_gate_destroy(gate);
        }
        double mean = 0;
        for (size_t i = 0; i < call_ms.size(); ++i) mean += call_ms[i] + drop_ms[i];
        mean /= (double)std::max<size_t>(call_ms.size(), 1);
        std::printf("{\"dropin_cpp\": {\"frame\": \"%dx%d, %d octaves x %d DoGs\", \"iterations\": %d, \"keypoints_per_frame\": %zu, "
                    "\"with_descriptor\": %zu, \"calculate_ms_median\": %.4f, \"drop_result_ms_median\": %.4f, \"ms_per_frame\": %.4f, "
                    "\"keypoints_per_s\": %.1f, \"two_gated_objects_ms_per_frame\": %.4f, \"two_gated_objects_keypoints_per_s\": %.1f, "
                    "\"abi_calculate_batch_ms\": %.4f, \"abi_sparse_fetch_ms\": %.4f}}\n",
                    w, h, octaves, dogs, iters, points, with_desc, median(call_ms), median(drop_ms), mean,
                    mean > 0 ? 1e3 * (double)points / mean : 0.0, pair_ms / (2.0 * iters), pair_ms > 0 ? 1e3 * (double)pair_points / pair_ms : 0.0,
                    abi_calc_ms, abi_fetch_ms);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}
