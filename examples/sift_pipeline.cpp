// Two sift::Sift objects on two host threads, joined by a gate (include/sift_hip.h): frames are handed out
// alternately, the device overlaps consecutive calculate() calls (the next frame's extrema pass runs under this
// frame's cleanup steps, pyramids never share the chip) and every result is the one a single object gives.
//   g++ -std=c++17 -pthread -Iinclude examples/sift_pipeline.cpp -Lsift_amd/lib -lsift_hip -Wl,-rpath,$PWD/sift_amd/lib -o sift_pipeline
//   GPU_MAX_HW_QUEUES=8 ./sift_pipeline tests/golden/parrot_r.pgm [frames=8]
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>

#include "sift/sift.hpp"

static bool read_pgm(const char* path, sift::Image2f& img) {
    std::ifstream f(path, std::ios::binary);
    std::string magic;
    int w = 0, h = 0, maxv = 0;
    if (!(f >> magic >> w >> h >> maxv) || magic != "P5" || maxv != 255) return false;
    f.get();
    std::vector<unsigned char> buf((size_t)w * (size_t)h);
    f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size());
    img.reshape(w, h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) img(x, y) = (float)buf[(size_t)x + (size_t)y * (size_t)w];
    return (bool)f;
}

static bool same(const std::vector<sift::InterestPoint>& a, const std::vector<sift::InterestPoint>& b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (a[i].loc.x != b[i].loc.x || a[i].loc.y != b[i].loc.y || a[i].scale != b[i].scale ||
            a[i].orientation != b[i].orientation || a[i].descriptors != b[i].descriptors)
            return false;
    return true;
}

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);   // before the first HIP call: contexts side by side want hardware queues of their own
    if (argc < 2) {
        std::cerr << "usage: " << argv[0] << " image.pgm [frames]\n";
        return 1;
    }
    const int frames = argc > 2 ? std::atoi(argv[2]) : 8;
    try {
        sift::Image2f base;
        if (!read_pgm(argv[1], base)) throw std::runtime_error("cannot read P5 PGM");
        // frame k: the image with its first k rows brightened, so that results differ from frame to frame
        std::vector<sift::Image2f> in((size_t)frames, base);
        for (int k = 0; k < frames; ++k)
            for (int y = 0; y < k && y < (int)base.height(); ++y)
                for (int x = 0; x < (int)base.width(); ++x) in[(size_t)k](x, y) = base(x, y) * 0.5f + 64.0f;

        std::vector<std::vector<sift::InterestPoint>> want((size_t)frames), got((size_t)frames);
        {
            sift::Sift single(3, 4);
            for (int k = 0; k < frames; ++k) {
                sift::Image2f img = in[(size_t)k];
                want[(size_t)k] = single.calculate(img);
            }
        }
        sift_hip_gate* gate = nullptr;
        if (sift_hip_gate_create(0, &gate) != SIFT_HIP_OK) throw std::runtime_error("sift_hip_gate_create failed");
        {
            sift::Sift a(3, 4), b(3, 4);
            a.join(gate);
            b.join(gate);
            std::atomic<int> next{0};
            auto worker = [&](sift::Sift& s) {
                for (int k = next++; k < frames; k = next++) {
                    sift::Image2f img = in[(size_t)k];
                    got[(size_t)k] = s.calculate(img);
                }
            };
            std::thread ta(worker, std::ref(a)), tb(worker, std::ref(b));
            ta.join();
            tb.join();
            a.join(nullptr);
            b.join(nullptr);
        }
        sift_hip_gate_destroy(gate);
        size_t total = 0;
        for (int k = 0; k < frames; ++k) {
            if (!same(want[(size_t)k], got[(size_t)k])) {
                std::cerr << "frame " << k << " differs\n";
                return 2;
            }
            total += got[(size_t)k].size();
        }
        std::cout << "ok: " << frames << " frames, " << total << " interest points, gated pair == single object\n";
    } catch (std::exception& ex) {
        std::cerr << ex.what() << std::endl;
        return 1;
    }
    return 0;
}
