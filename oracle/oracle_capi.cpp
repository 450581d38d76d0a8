// ORACLE — TEST INFRASTRUCTURE ONLY (see vigra_restate.hpp header).  Parity pinned against the reference's own
// prebuilt binary (oracle/refexec, tests/test_ref_pins.py).
// extern "C" surface so tests/ and bench.py's cpu_baseline leg can drive the oracle via ctypes.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <fstream>
#include <memory>

#include "sift_oracle.hpp"

using namespace oracle;

extern "C" {

struct oracle_params {
    uint16_t dogs_per_epoch;
    uint16_t octaves;
    float sigma;
    float k;
    uint8_t subpixel;
};

struct oracle_point {
    float scale;
    float orientation;
    uint16_t x, y;
    uint16_t octave, index;
    uint32_t filtered;
    int32_t cand_id;
    int32_t n_desc;
};

struct oracle_handle {
    std::unique_ptr<Sift> sift;
    Img image;  // the caller's image after calculate() (replaced when subpixel)
    std::vector<InterestPoint> result;
    int status = 0;  // 0 ok, 1 PreconditionViolation, 2 assertion
    double seconds = 0;
};

static void set_err(char* err, int errlen, const char* msg) {
    if (err && errlen > 0) {
        std::strncpy(err, msg, (size_t)errlen - 1);
        err[errlen - 1] = 0;
    }
}

static Params to_params(const oracle_params* p) {
    Params q;
    q.dogsPerEpoch = p->dogs_per_epoch;
    q.octaves = p->octaves;
    q.sigma = p->sigma;
    q.k = p->k;
    q.subpixel = p->subpixel != 0;
    return q;
}

// Runs Sift::calculate.  Always returns a handle; status tells how it ended.  On an exception the
// pyramid built so far and the (possibly replaced) image stay inspectable, like the reference's
// object state after a throw.
oracle_handle* oracle_run(const float* img, int w, int h, const oracle_params* p, int faithful,
                          char* err, int errlen) {
    auto* hd = new oracle_handle();
    hd->sift.reset(new Sift(to_params(p), faithful != 0));
    hd->image = Img(w, h);
    std::memcpy(hd->image.d.data(), img, sizeof(float) * (size_t)w * (size_t)h);
    set_err(err, errlen, "");
    const auto t0 = std::chrono::steady_clock::now();
    try {
        hd->result = hd->sift->calculate(hd->image);
    } catch (const PreconditionViolation& e) {
        hd->status = 1;
        set_err(err, errlen, e.what());
    } catch (const std::invalid_argument& e) {
        hd->status = 2;
        set_err(err, errlen, e.what());
    }
    hd->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return hd;
}
int oracle_status(const oracle_handle* h) { return h->status; }
double oracle_seconds(const oracle_handle* h) { return h->seconds; }
void oracle_free(oracle_handle* h) { delete h; }

void oracle_image_dims(const oracle_handle* h, int* w, int* ht) {
    *w = (int)h->image.w;
    *ht = (int)h->image.h;
}
void oracle_image_copy(const oracle_handle* h, float* out) {
    std::memcpy(out, h->image.d.data(), sizeof(float) * h->image.d.size());
}

// kind: 0 gaussian, 1 dog, 2 magnitude, 3 orientation (2/3: final state, i.e. AFTER the
// descriptor stage's in-place mutation; absent levels report 0x0)
static const Img* level(const oracle_handle* h, int kind, int o, int i) {
    const Sift& s = *h->sift;
    if (o < 0 || o >= s.octaves()) return nullptr;
    const int nl = (kind == 1) ? s.levels() - 1 : s.levels();
    if (i < 0 || i >= nl) return nullptr;
    switch (kind) {
        case 0: return &s.gaussian(o, i).img;
        case 1: return &s.dog(o, i).img;
        case 2: return s.magnitude(o, i);
        case 3: return s.orientation(o, i);
    }
    return nullptr;
}
int oracle_level_dims(const oracle_handle* h, int kind, int o, int i, int* w, int* ht) {
    const Img* g = level(h, kind, o, i);
    *w = g ? (int)g->w : 0;
    *ht = g ? (int)g->h : 0;
    return g != nullptr;
}
int oracle_level_copy(const oracle_handle* h, int kind, int o, int i, float* out) {
    const Img* g = level(h, kind, o, i);
    if (!g) return 0;
    std::memcpy(out, g->d.data(), sizeof(float) * g->d.size());
    return 1;
}
float oracle_level_scale(const oracle_handle* h, int kind, int o, int i) {
    return kind == 1 ? h->sift->dog(o, i).scale : h->sift->gaussian(o, i).scale;
}

// stage: 0 candidates (flags set), 1 after first cleanup, 2 after orientation assignment,
//        3 after second cleanup, 4 final (descriptors)
static const std::vector<InterestPoint>& stage_vec(const oracle_handle* h, int stage) {
    switch (stage) {
        case 0: return h->sift->trace.candidates;
        case 1: return h->sift->trace.after_sort1;
        case 2: return h->sift->trace.after_orient;
        case 3: return h->sift->trace.after_sort2;
        default: return h->result;
    }
}
int oracle_points_count(const oracle_handle* h, int stage) { return (int)stage_vec(h, stage).size(); }
void oracle_points_copy(const oracle_handle* h, int stage, oracle_point* out, float* desc) {
    const auto& v = stage_vec(h, stage);
    for (size_t n = 0; n < v.size(); ++n) {
        const InterestPoint& p = v[n];
        out[n].scale = p.scale;
        out[n].orientation = p.orientation;
        out[n].x = p.x; out[n].y = p.y;
        out[n].octave = p.octave; out[n].index = p.index;
        out[n].filtered = p.filtered ? 1u : 0u;
        out[n].cand_id = p.cand_id;
        out[n].n_desc = (int32_t)p.descriptors.size();
        if (desc) {
            float* d = desc + n * 128;
            for (int t = 0; t < 128; ++t) d[t] = t < (int)p.descriptors.size() ? p.descriptors[(size_t)t] : 0.0f;
        }
    }
}

// The result file of the reference's command line program (main.cpp:78-89), written with the same iostream inserters
// (operator<< of u16_t and f32_t at the default precision): the expected text of the CLI parity tests.
int oracle_write_result(const oracle_handle* h, const char* path) {
    std::ofstream out(path);
    if (!out) return 1;
    out << "Location\tscale\torientation\tdescriptors\n";
    for (const InterestPoint& p : h->result) {
        out << "[" << p.x << ", " << p.y << "]\t" << p.scale << "\t" << p.orientation << "\t" << "[";
        for (f32 d : p.descriptors) {
            out << d << ", ";
        }
        out << "]\n";
    }
    out.close();
    return out ? 0 : 1;
}

// ---- known-answer-test helpers -------------------------------------------------------------
int oracle_gauss_taps(float sigma, float* taps, int cap) {
    const Kernel1D k = initGaussian((double)sigma);
    for (int i = 0; i < (int)k.k.size() && i < cap; ++i) taps[i] = k.k[(size_t)i];
    return k.radius;
}
int oracle_convolve(const float* in, int w, int h, float sigma, float* out, char* err, int errlen) {
    Img a(w, h);
    std::memcpy(a.d.data(), in, sizeof(float) * a.d.size());
    try {
        const Img r = convolveWithGauss(a, sigma);
        std::memcpy(out, r.d.data(), sizeof(float) * r.d.size());
    } catch (const PreconditionViolation& e) {
        set_err(err, errlen, e.what());
        return 1;
    }
    return 0;
}
void oracle_resize_index_map(int wold, int wnew, int* out) {
    const auto m = resizeIndexMap(wold, wnew);
    std::copy(m.begin(), m.end(), out);
}
// mode 0: reduceToNextLevel, 1: increaseToNextLevel; out must hold the new size
int oracle_resample(const float* in, int w, int h, float sigma, int mode, float* out, char* err,
                    int errlen) {
    Img a(w, h);
    std::memcpy(a.d.data(), in, sizeof(float) * a.d.size());
    try {
        const Img r = mode == 0 ? reduceToNextLevel(a, sigma) : increaseToNextLevel(a, sigma);
        std::memcpy(out, r.d.data(), sizeof(float) * r.d.size());
    } catch (const PreconditionViolation& e) {
        set_err(err, errlen, e.what());
        return 1;
    }
    return 0;
}
void oracle_dog(const float* lower, const float* higher, int w, int h, float* out) {
    Img a(w, h), b(w, h);
    std::memcpy(a.d.data(), lower, sizeof(float) * a.d.size());
    std::memcpy(b.d.data(), higher, sizeof(float) * b.d.size());
    const Img r = dog(a, b);
    std::memcpy(out, r.d.data(), sizeof(float) * r.d.size());
}
float oracle_vertex_parabola(uint16_t lnx, float lny, uint16_t px, float py, uint16_t rnx, float rny) {
    return vertexParabola(lnx, lny, px, py, rnx, rny);
}
int oracle_edge_filtered(const float* d0, const float* d1, const float* d2, int w, int h, int x, int y) {
    Img a(w, h), b(w, h), c(w, h);
    std::memcpy(a.d.data(), d0, sizeof(float) * a.d.size());
    std::memcpy(b.d.data(), d1, sizeof(float) * b.d.size());
    std::memcpy(c.d.data(), d2, sizeof(float) * c.d.size());
    const Img* p[3] = {&a, &b, &c};
    return edgeResponseFiltered(p, x, y) ? 1 : 0;
}
void oracle_gradient(const float* in, int w, int h, float* mag, float* ori) {
    Img a(w, h);
    std::memcpy(a.d.data(), in, sizeof(float) * a.d.size());
    std::memset(mag, 0, sizeof(float) * a.d.size());
    std::memset(ori, 0, sizeof(float) * a.d.size());
    for (long x = 1; x < w - 1; ++x)
        for (long y = 1; y < h - 1; ++y) {
            mag[x + y * w] = gradientMagnitude(a, x, y);
            ori[x + y * w] = gradientOrientation(a, x, y);
        }
}
// 3x3 inverse + solve exposed for linear-algebra KATs (row-fastest storage like vigra::Matrix)
int oracle_inverse3(const float* a, float* res) {
    Mat A(3, 3), R(3, 3);
    std::memcpy(A.d.data(), a, sizeof(float) * 9);
    const bool ok = inverse(A.v(), R.v());
    std::memcpy(res, R.d.data(), sizeof(float) * 9);
    return ok ? 1 : 0;
}
int oracle_solve3(const float* a, const float* b, float* res) {
    Mat A(3, 3), B(3, 1), R(3, 1);
    std::memcpy(A.d.data(), a, sizeof(float) * 9);
    std::memcpy(B.d.data(), b, sizeof(float) * 3);
    const bool ok = linearSolve(A.v(), B.v(), R.v());
    std::memcpy(res, R.d.data(), sizeof(float) * 3);
    return ok ? 1 : 0;
}
// libstdc++ std::sort with InterestPoint::cmpByFilter on full-size elements; perm[i] = original
// index of the element that ends at position i.  Used to validate the product's cleanup stage.
void oracle_sort_by_filter(const uint8_t* flags, int n, int32_t* perm) {
    std::vector<InterestPoint> v((size_t)n);
    for (int i = 0; i < n; ++i) {
        v[(size_t)i].filtered = flags[i] != 0;
        v[(size_t)i].cand_id = i;
    }
    std::sort(v.begin(), v.end(), InterestPoint::cmpByFilter);
    for (int i = 0; i < n; ++i) perm[i] = v[(size_t)i].cand_id;
}
float oracle_atan2f(float y, float x) { return std::atan2(y, x); }
uint16_t oracle_f32_to_u16(float v) { return f32_to_u16_x86(v); }

}  // extern "C"
