// Minimal C++ user of the drop-in API: reads a binary PGM (P5), runs sift::Sift::calculate() on the
// GPU and writes the reference's result file format (/root/reference/main.cpp:78-89).
//   g++ -std=c++17 -Iinclude examples/sift_points.cpp -Lsift_amd/lib -lsift_hip -Wl,-rpath,$PWD/sift_amd/lib -o sift_points
//   ./sift_points tests/golden/parrot_r.pgm [octaves=4] [dogsPerEpoch=3] [subpixel=0]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

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

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cerr << "usage: " << argv[0] << " image.pgm [octaves] [dogsPerEpoch] [subpixel]\n";
        return 1;
    }
    const u16_t octaves = argc > 2 ? (u16_t)std::atoi(argv[2]) : 4;
    const u16_t dogs = argc > 3 ? (u16_t)std::atoi(argv[3]) : 3;
    const bool subpixel = argc > 4 && std::atoi(argv[4]) != 0;
    try {
        sift::Image2f img;
        if (!read_pgm(argv[1], img)) throw std::runtime_error("cannot read P5 PGM");
        sift::Sift sift(dogs, octaves, 1.6f, std::sqrt(2.0f), subpixel);
        std::vector<sift::InterestPoint> interestPoints = sift.calculate(img);
        std::ofstream out("interstpoints.txt");
        out << "Location\tscale\torientation\tdescriptors\n";
        for (const sift::InterestPoint& p : interestPoints) {
            out << "[" << p.loc.x << ", " << p.loc.y << "]\t" << p.scale << "\t" << p.orientation << "\t[";
            for (f32_t d : p.descriptors) out << d << ", ";
            out << "]\n";
        }
        std::cout << interestPoints.size() << " interest points -> interstpoints.txt\n";
    } catch (std::exception& ex) {
        std::cerr << ex.what() << std::endl;
    }
    return 0;
}
