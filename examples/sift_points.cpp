// C++ user of the drop-in API shaped like the reference's main() (/root/reference/main.cpp:47-92): image ingest (band 0 of
// a PGM / PPM / PNG / JPEG, main.cpp:52-54), sift::Sift::calculate() on the GPU, the overlay <img>_orientation.png (:59-76) and the
// result file interstpoints.txt (:78-89).  Everything the reference gets from Vigra impex / OpenCV comes from the C ABI.
//   g++ -std=c++17 -Iinclude examples/sift_points.cpp -Lsift_amd/lib -lsift_hip -Wl,-rpath,$PWD/sift_amd/lib -o sift_points
//   ./sift_points tests/golden/parrot_r.pgm [octaves=4] [dogsPerEpoch=3] [subpixel=0]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "sift/sift.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cerr << "usage: " << argv[0] << " image.{pgm,ppm,png} [octaves] [dogsPerEpoch] [subpixel]\n";
        return 1;
    }
    const std::string img_file = argv[1];
    const u16_t octaves = argc > 2 ? (u16_t)std::atoi(argv[2]) : 4;
    const u16_t dogsPerEpoch = argc > 3 ? (u16_t)std::atoi(argv[3]) : 3;
    const bool subpixel = argc > 4 && std::atoi(argv[4]) != 0;
    try {
        char err[512] = "";
        int w = 0, h = 0;
        int bits = 0;
        if (sift_hip_image_info(img_file.c_str(), &w, &h, nullptr, &bits, err, sizeof(err)) != SIFT_HIP_OK) throw std::runtime_error(err);
        sift::Image2f img(w, h);
        if (sift_hip_image_read_band0(img_file.c_str(), img.data(), (long long)w * h, err, sizeof(err)) != SIFT_HIP_OK) throw std::runtime_error(err);

        sift::Sift sift(dogsPerEpoch, octaves, 1.6f, std::sqrt(2.0f), subpixel);
        std::vector<sift::InterestPoint> interestPoints;
        if (bits == 8) {   // an 8-bit file: its samples go to the GPU as bytes and are widened there (same floats, same result)
            std::vector<unsigned char> px((size_t)w * (size_t)h);
            for (size_t i = 0; i < px.size(); ++i) px[i] = (unsigned char)img.data()[i];
            interestPoints = sift.calculate(px.data(), w, h);
        } else {
            interestPoints = sift.calculate(img);
        }

        // main.cpp:59-76: boxes on the colour image
        std::vector<uint8_t> image((size_t)w * (size_t)h * 3);
        if (sift_hip_image_read_bgr8(img_file.c_str(), image.data(), (long long)image.size(), err, sizeof(err)) != SIFT_HIP_OK) throw std::runtime_error(err);
        std::vector<sift_hip_keypoint> boxes(interestPoints.size());
        for (size_t i = 0; i < boxes.size(); ++i) {
            const sift::InterestPoint& p = interestPoints[i];
            boxes[i] = sift_hip_keypoint{p.scale, p.orientation, p.loc.x, p.loc.y, p.octave, p.index, (uint8_t)p.filtered, (uint8_t)!p.descriptors.empty(), 0};
        }
        sift_hip_overlay_draw(image.data(), w, h, boxes.data(), (long long)boxes.size(), sift.subpixel ? 1 : 0);
        if (sift_hip_png_write_bgr8((img_file + "_orientation.png").c_str(), image.data(), w, h, err, sizeof(err)) != SIFT_HIP_OK) throw std::runtime_error(err);

        // main.cpp:78-89
        std::ofstream out("interstpoints.txt");
        out << "Location\tscale\torientation\tdescriptors\n";
        for (const sift::InterestPoint& p : interestPoints) {
            out << "[" << p.loc.x << ", " << p.loc.y << "]\t" << p.scale << "\t" << p.orientation << "\t[";
            for (f32_t d : p.descriptors) out << d << ", ";
            out << "]\n";
        }
        out.close();
        std::cout << interestPoints.size() << " interest points -> interstpoints.txt\n";
    } catch (std::exception& ex) {
        std::cerr << ex.what() << std::endl;
    }
    return 0;
}
