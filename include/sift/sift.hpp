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
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
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
        // the lists come down in the sparse format (collect): let the descriptor kernel count its floats while they are in
        // registers, so that the size is known when the batch ends (no counting pass, no second wait)
        (void)sift_hip_set_option(_ctx, "wire_count", 1);
    }
    Sift(const Sift&) = delete;
    Sift& operator=(const Sift&) = delete;
    ~Sift() {
        stop_workers();
        sift_hip_destroy(_ctx);
    }

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
    //
    // The reference's return type fixes one heap block per keypoint (`std::vector<f32_t> descriptors`, interestpoint.hpp:46):
    // what is left to this function is everything else.  The transfer buffers live in the object (a fresh 10 MB vector per call
    // is ~2500 page faults), no dense n x 128 array is built on the way (a block of 64 descriptors is expanded into a scratch
    // that stays in L1 and handed to the points at once), and a few threads share the points of a large result - each its own
    // contiguous range, so every thread allocates from its own malloc arena.
    static int mask_popcount(const unsigned char* m) {   // set bits of a record's 14 presence bytes
        unsigned long long a, b;
        std::memcpy(&a, m, 8);
        std::memcpy(&b, m + 6, 8);
        return __builtin_popcountll(a) + __builtin_popcountll(b >> 16);
    }
    void expand_range(std::vector<InterestPoint>& out, size_t i0, size_t i1, const float* val) const {
        constexpr size_t kBlock = 64;
        sift_hip_keypoint kp[kBlock];
        float desc[kBlock * 128];
        for (size_t i = i0; i < i1; i += kBlock) {
            const size_t m = i1 - i < kBlock ? i1 - i : kBlock;
            const unsigned char* r = _rec.data() + i * 34;
            size_t nv = 0;
            for (size_t k = 0; k < m; ++k) nv += (size_t)mask_popcount(r + k * 34 + 20);
            // (the library's vector form looks 8 floats ahead; _val keeps 8 floats of slack behind the last value)
            if (sift_hip_sparse_unpack_host(r, val, (int64_t)m, kp, desc, 1) != SIFT_HIP_OK) throw std::runtime_error("sift_hip_sparse_unpack_host failed");
            for (size_t k = 0; k < m; ++k) {
                fill(out[i + k], kp[k]);
                if (kp[k].has_descriptor) out[i + k].descriptors.assign(desc + k * 128, desc + (k + 1) * 128);
            }
            val += nv;
        }
    }
    std::vector<InterestPoint> collect() {
        const long long n = sift_hip_result_total(_ctx);
        std::vector<InterestPoint> out((size_t)(n > 0 ? n : 0));
        if (n <= 0) return out;
        int64_t nnz = 0;
        int lossless = 0;
        if (sift_hip_result_sparse_size(_ctx, &nnz, &lossless) == SIFT_HIP_OK && lossless) {
            if (_rec.size() < (size_t)n * 34) _rec.resize((size_t)n * 34);
            if (_val.size() < (size_t)nnz + 8) _val.resize((size_t)nnz + 8);
            if (sift_hip_result_copy_sparse(_ctx, _rec.data(), _val.data()) != SIFT_HIP_OK) throw std::runtime_error("sift_hip_result_copy_sparse failed");
            unsigned hw = std::thread::hardware_concurrency();
            const size_t threads = (size_t)n < 4096 ? 1 : (hw >= 8 ? 4 : (hw >= 4 ? 2 : 1));
            if (threads == 1) {
                expand_range(out, 0, (size_t)n, _val.data());
                return out;
            }
            // first value of every thread's range: the presence bits say how many floats the points in front of it own
            const size_t per = (((size_t)n + threads - 1) / threads + 63) / 64 * 64;
            std::vector<size_t> first(threads, 0);
            {
                size_t v = 0;
                for (size_t i = 0; i < (size_t)n; ++i) {
                    if (i % per == 0) first[i / per] = v;
                    v += (size_t)mask_popcount(_rec.data() + i * 34 + 20);
                }
            }
            // the object's worker threads (started at the first large result, parked on a condition variable between calls:
            // starting threads per call costs as much as a worker's share of the expansion)
            std::vector<std::exception_ptr> failed(threads);
            auto part = [&](size_t t) {
                const size_t i0 = t * per < (size_t)n ? t * per : (size_t)n, i1 = i0 + per < (size_t)n ? i0 + per : (size_t)n;
                try { if (i0 < i1) expand_range(out, i0, i1, _val.data() + first[t]); } catch (...) { failed[t] = std::current_exception(); }
            };
            run_on_workers(threads - 1, [&](size_t w) { part(w + 1); });
            part(0);
            wait_workers();
            for (auto& f : failed) if (f) std::rethrow_exception(f);
            return out;
        }
        if (_kp.size() < (size_t)n) _kp.resize((size_t)n);
        if (_desc.size() < (size_t)n * 128) _desc.resize((size_t)n * 128);
        if (sift_hip_result_copy(_ctx, _kp.data(), _desc.data()) != SIFT_HIP_OK) throw std::runtime_error("sift_hip_result_copy failed");
        for (size_t i = 0; i < out.size(); ++i) {
            fill(out[i], _kp[i]);
            if (_kp[i].has_descriptor) out[i].descriptors.assign(_desc.begin() + (std::ptrdiff_t)i * 128, _desc.begin() + (std::ptrdiff_t)(i + 1) * 128);
        }
        return out;
    }
    // ---- worker threads of collect(): a job is a function of the worker's index; run_on_workers hands the first `count` workers
    // the job, wait_workers returns when they are done
    std::vector<std::thread> _workers;
    std::mutex _wm;
    std::condition_variable _wcv;
    std::function<void(size_t)> _job;
    size_t _job_workers = 0, _job_done = 0;
    unsigned long long _job_id = 0;
    bool _stop = false;
    void worker_main(size_t index) {
        unsigned long long seen = 0;
        for (;;) {
            std::function<void(size_t)> job;
            {
                std::unique_lock<std::mutex> lk(_wm);
                _wcv.wait(lk, [&] { return _stop || (_job_id != seen && index < _job_workers); });
                if (_stop) return;
                seen = _job_id;
                job = _job;
            }
            job(index);
            {
                std::lock_guard<std::mutex> lk(_wm);
                ++_job_done;
            }
            _wcv.notify_all();
        }
    }
    void run_on_workers(size_t count, std::function<void(size_t)> job) {
        while (_workers.size() < count) {
            const size_t index = _workers.size();
            _workers.emplace_back([this, index] { worker_main(index); });
        }
        {
            std::lock_guard<std::mutex> lk(_wm);
            _job = std::move(job);
            _job_workers = count;
            _job_done = 0;
            ++_job_id;
        }
        _wcv.notify_all();
    }
    void wait_workers() {
        std::unique_lock<std::mutex> lk(_wm);
        _wcv.wait(lk, [&] { return _job_done == _job_workers; });
        _job_workers = 0;
    }
    void stop_workers() {
        {
            std::lock_guard<std::mutex> lk(_wm);
            _stop = true;
        }
        _wcv.notify_all();
        for (auto& t : _workers) t.join();
        _workers.clear();
    }
    // transfer buffers, kept between calls (collect)
    std::vector<unsigned char> _rec;
    std::vector<float> _val;
    std::vector<sift_hip_keypoint> _kp;
    std::vector<float> _desc;
};

}  // namespace sift
#endif
