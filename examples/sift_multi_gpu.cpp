// A batch of frames over every GPU of the node from ONE C++ process, no Python, no MPI: the layout of SURVEY.md 8(e) through
// the C ABI (sift_hip_group_*: one context and host thread per GPU, contiguous blocks of frames, keypoint lists gathered
// over RCCL - or device-to-device copies where a GPU is listed twice - on the first GPU in global image order, two batches in
// flight: sift_hip_group_submit / _collect).  Each frame also goes through a plain sift::Sift object and the two results are
// compared, so the program doubles as a check.
//   g++ -std=c++17 -pthread -Iinclude examples/sift_multi_gpu.cpp -Lsift_amd/lib -lsift_hip -Wl,-rpath,$PWD/sift_amd/lib -o sift_multi_gpu
//   ./sift_multi_gpu image.pgm [frames=16] [shards=number of GPUs]        (shards > GPUs: several shards per GPU)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "sift/sift.hpp"

extern "C" int hipGetDeviceCount(int*);

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);   // before the first HIP call: contexts side by side want hardware queues of their own
    if (argc < 2) {
        std::cerr << "usage: " << argv[0] << " image.{pgm,ppm,png} [frames] [shards]\n";
        return 1;
    }
    const int frames = argc > 2 ? std::atoi(argv[2]) : 16;
    int gpus = 0;
    if (hipGetDeviceCount(&gpus) != 0 || gpus < 1) { std::cerr << "no GPU\n"; return 1; }
    const int shards = argc > 3 ? std::atoi(argv[3]) : gpus;
    char err[512] = "";
    int w = 0, h = 0;
    if (sift_hip_image_info(argv[1], &w, &h, nullptr, nullptr, err, sizeof(err)) != SIFT_HIP_OK) { std::cerr << err << "\n"; return 1; }
    std::vector<float> base((size_t)w * h);
    if (sift_hip_image_read_band0(argv[1], base.data(), (long long)base.size(), err, sizeof(err)) != SIFT_HIP_OK) { std::cerr << err << "\n"; return 1; }
    // three batches of frames that differ: the image shifted cyclically by a few columns each
    const int kBatches = 3;
    std::vector<std::vector<float>> batch(kBatches, std::vector<float>((size_t)frames * base.size()));
    for (int b = 0; b < kBatches; ++b)
        for (int f = 0; f < frames; ++f)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x)
                    batch[(size_t)b][(size_t)f * base.size() + (size_t)y * w + x] = base[(size_t)y * w + (size_t)((x + 7 * f + 3 * b) % w)];

    std::vector<int> devices(shards);
    for (int s = 0; s < shards; ++s) devices[s] = s % gpus;
    sift_hip_group* g = nullptr;
    if (sift_hip_group_create(devices.data(), shards, &g, err, sizeof(err)) != SIFT_HIP_OK) { std::cerr << err << "\n"; return 1; }
    char how[256] = "";
    const int over_rccl = sift_hip_group_transport(g, how, sizeof(how));
    sift_hip_params p{};
    p.dogs_per_epoch = 3; p.octaves = 4; p.sigma = 1.6f; p.k = std::sqrt(2.0f); p.subpixel = 0;
    sift::Sift single(3, 4, 1.6f, std::sqrt(2.0f), false, 0);
    long long all = 0;
    double cms = 0, gms = 0, xms = 0;
    int64_t gb = 0;
    // two batches in flight: batch b+1 is submitted before batch b is collected, so b's gather runs under b+1's kernels
    if (sift_hip_group_submit(g, batch[0].data(), frames, w, h, &p, err, sizeof(err)) != SIFT_HIP_OK) { std::cerr << "submit: " << err << "\n"; return 1; }
    for (int b = 0; b < kBatches; ++b) {
        if (b + 1 < kBatches && sift_hip_group_submit(g, batch[(size_t)b + 1].data(), frames, w, h, &p, err, sizeof(err)) != SIFT_HIP_OK) {
            std::cerr << "submit: " << err << "\n";
            return 1;
        }
        if (sift_hip_group_collect(g, err, sizeof(err)) != SIFT_HIP_OK) { std::cerr << "collect: " << err << "\n"; return 1; }
        const long long total = sift_hip_group_result_total(g);
        std::vector<int32_t> counts(frames);
        sift_hip_group_result_counts(g, counts.data(), frames);
        std::vector<sift_hip_keypoint> kp((size_t)total);
        std::vector<float> desc((size_t)total * 128);
        sift_hip_group_result_copy(g, kp.data(), desc.data());
        double c1 = 0, g1 = 0, x1 = 0;
        sift_hip_group_timing(g, &c1, &g1, &gb);
        sift_hip_group_gather_exposed(g, &x1);
        cms += c1; gms += g1; xms += x1;
        all += total;
        // the same frames one by one through the drop-in class
        long long at = 0;
        for (int f = 0; f < frames && !std::getenv("SIFT_EXAMPLE_NOCHECK"); ++f) {
            sift::Image2f img(w, h);
            std::memcpy(img.data(), batch[(size_t)b].data() + (size_t)f * base.size(), base.size() * sizeof(float));
            const std::vector<sift::InterestPoint> pts = single.calculate(img);
            if ((long long)pts.size() != counts[f]) { std::cerr << "batch " << b << " frame " << f << ": " << pts.size() << " points vs " << counts[f] << "\n"; return 2; }
            for (size_t i = 0; i < pts.size(); ++i, ++at) {
                const sift_hip_keypoint& k = kp[(size_t)at];
                if (k.x != pts[i].loc.x || k.y != pts[i].loc.y || k.octave != pts[i].octave ||
                    std::memcmp(&k.orientation, &pts[i].orientation, 4) != 0 ||
                    (pts[i].descriptors.size() == 128 && std::memcmp(pts[i].descriptors.data(), &desc[(size_t)at * 128], 512) != 0)) {
                    std::cerr << "batch " << b << " frame " << f << " point " << i << " differs\n";
                    return 2;
                }
            }
        }
    }
    std::printf("ok: %d frames over %d shards on %d GPU(s), %d batches in a pipeline, %lld keypoints; gather over %s (%s); per batch: shards %.2f ms, "
                "gather %.2f ms of which %.2f ms exposed, %lld bytes across devices\n", frames, shards, gpus, kBatches, all,
                over_rccl ? "RCCL" : "copies", how, cms / kBatches, gms / kBatches, xms / kBatches, (long long)gb);
    std::fflush(stdout);
    sift_hip_group_destroy(g);
    return 0;
}
