// sift::Sift — drop-in for the reference's class (/root/reference/sift.hpp:17-78): same public
// member `subpixel`, same constructor arguments and defaults, same `calculate(image&)` returning
// std::vector<InterestPoint> by value, same side effect on the caller's image when subpixel
// (sift.cpp:20-21), exceptions with Vigra's message text where the reference throws
// vigra::PreconditionViolation.  Header-only host code above the C ABI of libsift_hip.so
// (include/sift_hip.h); every stage runs in HIP kernels on the GPU.
#ifndef SIFT_AMD_SIFT_HPP
#define SIFT_AMD_SIFT_HPP
#include <cassert>
#include <cmath>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../sift_hip.h"
#include "image2f.hpp"
#include "interestpoint.hpp"
#include "types.hpp"
#ifdef SIFT_WITH_VIGRA
#include <vigra/multi_array.hxx>
#endif

namespace sift {

// vigra::PreconditionViolation analogue (also a std::exception); what() carries Vigra's text.
class PreconditionViolation : public std::runtime_error {
public:
    explicit PreconditionViolation(const std::string& m) : std::runtime_error(m) {}
};

class Sift {
public:
    const bool subpixel;

    explicit Sift(u16_t dogsPerEpoch = 3, u16_t octaves = 3, f32_t sigma = 1.6, f32_t k = std::sqrt(2),
                  bool subpixel_ = false, int device = 0)
        : subpixel(subpixel_), _sigma(sigma), _k(k), _dogsPerEpoch(dogsPerEpoch), _octaves(octaves) {
        char err[512] = "";
        if (sift_hip_create(device, &_ctx, err, sizeof(err)) != SIFT_HIP_OK)
            throw std::runtime_error(std::string("sift_hip_create: ") + err);
    }
    Sift(const Sift&) = delete;
    Sift& operator=(const Sift&) = delete;
    ~Sift() { sift_hip_destroy(_ctx); }

    std::vector<InterestPoint> calculate(Image2f& img) {
        const int w = (int)img.width(), h = (int)img.height();
        const int rc = run(img.data(), w, h);
        if (subpixel) {  // the reference overwrites the caller's image before anything can throw
            int nw = 0, nh = 0;
            if (sift_hip_image_dims(_ctx, &nw, &nh) == SIFT_HIP_OK && (nw != w || nh != h)) {
                Image2f up(nw, nh);
                if (sift_hip_image_copy(_ctx, 0, up.data()) == SIFT_HIP_OK) img = up;
            }
        }
        raise(rc);
        return collect();
    }

    // Extension (not in the reference): an 8-bit frame as a host that has just decoded an 8-bit file holds it (the reference's
    // inputs are such files, main.cpp:52-54).  A quarter of the bytes cross the link; the GPU widens them to the integer-valued
    // floats vigra::importImage would have produced, so the result equals calculate() on (float)pixel.  The caller's pixels are
    // never replaced (with subpixel the upsampled image stays in the library: sift_hip_image_copy).
    std::vector<InterestPoint> calculate(const unsigned char* pixels, int w, int h) {
        sift_hip_params p = params();
        raise(sift_hip_calculate_batch_u8(_ctx, pixels, 1, w, h, &p, _err, sizeof(_err)));
        return collect();
    }

    // Extension (not in the reference): Sift objects on one GPU, one host thread each, can be joined by a gate
    // (sift_hip_gate_create) so that their calculate() calls overlap on the device while no pyramid shares the chip
    // (include/sift_hip.h).  nullptr detaches.
    void join(sift_hip_gate* gate) {
        if (sift_hip_set_gate(_ctx, gate) != SIFT_HIP_OK) throw std::invalid_argument("sift::Sift::join: gate of another device");
    }

#ifdef SIFT_WITH_VIGRA
    std::vector<InterestPoint> calculate(vigra::MultiArray<2, f32_t>& img) {
        const int w = (int)img.width(), h = (int)img.height();
        const int rc = run(img.data(), w, h);
        if (subpixel) {
            int nw = 0, nh = 0;
            if (sift_hip_image_dims(_ctx, &nw, &nh) == SIFT_HIP_OK && (nw != w || nh != h)) {
                vigra::MultiArray<2, f32_t> up(vigra::Shape2(nw, nh));
                if (sift_hip_image_copy(_ctx, 0, up.data()) == SIFT_HIP_OK) img = up;
            }
        }
        raise(rc);
        return collect();
    }
#endif

private:
    const f32_t _sigma, _k;
    const u16_t _dogsPerEpoch, _octaves;
    sift_hip_ctx* _ctx = nullptr;
    char _err[512] = "";

    sift_hip_params params() const {
        sift_hip_params p{};
        p.dogs_per_epoch = _dogsPerEpoch;
        p.octaves = _octaves;
        p.sigma = _sigma;
        p.k = _k;
        p.subpixel = subpixel ? 1 : 0;
        return p;
    }
    int run(const float* data, int w, int h) {
        const sift_hip_params p = params();
        return sift_hip_calculate_batch(_ctx, data, 1, w, h, &p, _err, sizeof(_err));
    }
    void raise(int rc) const {
        if (rc == SIFT_HIP_OK) return;
        if (rc == SIFT_HIP_EPRECONDITION) throw PreconditionViolation(_err);
        if (rc == SIFT_HIP_EASSERT) {  // the reference assert()s (sift.cpp:382-383)
            assert(!"sift::Sift: octaves > 0 && dogsPerEpoch >= 3");
            throw std::logic_error(_err);
        }
        throw std::runtime_error(_err);
    }
    static void fill(InterestPoint& p, const sift_hip_keypoint& k) {
        p.scale = k.scale;
        p.octave = k.octave;
        p.index = k.index;
        p.filtered = k.filtered != 0;
        p.loc = Point<u16_t, u16_t>(k.x, k.y);
        p.orientation = k.orientation;
    }
    // The keypoint lists come down in the library's sparse format (34-byte records = the 20-byte keypoint + 112 presence bits,
    // and only the descriptor floats that are not +0.0f: ~200 instead of 532 bytes per keypoint over the link) and are expanded
    // straight into the InterestPoints; results the format would lose (a bin 7 that is not +0.0f) take the dense arrays.
    std::vector<InterestPoint> collect() {
        const long long n = sift_hip_result_total(_ctx);
        std::vector<InterestPoint> out((size_t)(n > 0 ? n : 0));
        if (n <= 0) return out;
        int64_t nnz = 0;
        int lossless = 0;
        if (sift_hip_result_sparse_size(_ctx, &nnz, &lossless) == SIFT_HIP_OK && lossless) {
            std::vector<unsigned char> rec((size_t)n * 34);
            std::vector<float> val((size_t)(nnz > 0 ? nnz : 1));
            if (sift_hip_result_copy_sparse(_ctx, rec.data(), val.data()) != SIFT_HIP_OK) throw std::runtime_error("sift_hip_result_copy_sparse failed");
            // expanded by the library's host routine (vectorised, a few threads), then handed to the InterestPoints
            std::vector<sift_hip_keypoint> kp((size_t)n);
            std::vector<float> desc((size_t)n * 128);
            unsigned hw = std::thread::hardware_concurrency();
            if (sift_hip_sparse_unpack_host(rec.data(), val.data(), n, kp.data(), desc.data(), (int)(hw > 8 ? 8 : (hw ? hw : 1))) != SIFT_HIP_OK)
                throw std::runtime_error("sift_hip_sparse_unpack_host failed");
            for (size_t i = 0; i < out.size(); ++i) {
                fill(out[i], kp[i]);
                if (kp[i].has_descriptor) out[i].descriptors.assign(desc.begin() + (std::ptrdiff_t)i * 128, desc.begin() + (std::ptrdiff_t)(i + 1) * 128);
            }
            return out;
        }
        std::vector<sift_hip_keypoint> kp((size_t)n);
        std::vector<float> desc((size_t)n * 128);
        if (sift_hip_result_copy(_ctx, kp.data(), desc.data()) != SIFT_HIP_OK) throw std::runtime_error("sift_hip_result_copy failed");
        for (size_t i = 0; i < out.size(); ++i) {
            fill(out[i], kp[i]);
            if (kp[i].has_descriptor) out[i].descriptors.assign(desc.begin() + (std::ptrdiff_t)i * 128, desc.begin() + (std::ptrdiff_t)(i + 1) * 128);
        }
        return out;
    }
};

}  // namespace sift
#endif
