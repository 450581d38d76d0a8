// ORACLE — TEST INFRASTRUCTURE ONLY (see vigra_restate.hpp header).  Parity pinned against the reference's own
// prebuilt binary (oracle/refexec, tests/test_ref_pins.py).
// CPU restatement of /root/reference/sift.cpp + algorithms.cpp.  Compile with -ffp-contract=off.
#include "sift_oracle.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace oracle {

// ------------------------------------------------------------------------------------------
// sift::alg (algorithms.cpp)
// ------------------------------------------------------------------------------------------

// algorithms.cpp:24-36
Img reduceToNextLevel(const Img& img, f32 sigma) {
    const long w = (img.w + 1) / 2, h = (img.h + 1) / 2;
    return resizeImageNoInterpolation(convolveWithGauss(img, sigma), w, h);
}

// algorithms.cpp:38-49
Img increaseToNextLevel(const Img& img, f32 sigma) {
    return resizeImageNoInterpolation(convolveWithGauss(img, sigma), img.w * 2, img.h * 2);
}

// algorithms.cpp:52-64
Img dog(const Img& lower, const Img& higher) {
    Img result(lower.w, lower.h);
    for (long x = 0; x < lower.w; ++x)
        for (long y = 0; y < lower.h; ++y) {
            const f32 dif = higher(x, y) - lower(x, y);
            result(x, y) = 128.0f + dif;
        }
    return result;
}

// algorithms.cpp:66-77 (signs are "reversed" in the reference: I(x-1) - I(x+1))
void foDerivative(const Img* const img[3], long x, long y, f32 d[3]) {
    d[0] = ((*img[1])(x - 1, y) - (*img[1])(x + 1, y)) / 2.0f;
    d[1] = ((*img[1])(x, y - 1) - (*img[1])(x, y + 1)) / 2.0f;
    d[2] = ((*img[0])(x, y) - (*img[2])(x, y)) / 2.0f;
}

// algorithms.cpp:79-106.  h[i][j] = sec_deriv(i, j).  dys' first two terms cancel (:91-92).
void soDerivative(const Img* const img[3], long x, long y, f32 h[3][3]) {
    const Img &i0 = *img[0], &i1 = *img[1], &i2 = *img[2];
    const f32 dxx = i1(x + 1, y) + i1(x - 1, y) - 2.0f * i1(x, y);
    const f32 dyy = i1(x, y + 1) + i1(x, y - 1) - 2.0f * i1(x, y);
    const f32 dss = i2(x, y) + i0(x, y) - 2.0f * i1(x, y);
    const f32 dxy =
        (i1(x + 1, y + 1) - i1(x - 1, y + 1) - i1(x + 1, y - 1) + i1(x - 1, y - 1)) / 2.0f;
    const f32 dxs = (i2(x + 1, y) - i2(x - 1, y) - i0(x + 1, y) + i0(x - 1, y)) / 2.0f;
    const f32 dys = (i2(x, y + 1) - i2(x, y + 1) - i0(x, y + 1) + i0(x, y - 1)) / 2.0f;
    h[0][0] = dxx; h[1][0] = dxy; h[2][0] = dxs;
    h[0][1] = dxy; h[1][1] = dyy; h[2][1] = dys;
    h[0][2] = dxs; h[1][2] = dys; h[2][2] = dss;
}

// algorithms.cpp:108-111: std::pow(float,int) and std::sqrt run in double (bin@0x413890-0x4139f2)
f32 gradientMagnitude(const Img& img, long x, long y) {
    const f32 dx = img(x + 1, y) - img(x - 1, y);
    const f32 dy = img(x, y + 1) - img(x, y - 1);
    return (f32)std::sqrt((double)dx * (double)dx + (double)dy * (double)dy);
}

// algorithms.cpp:113-116: atan2f (radians!), float add of 360, fmod in double
f32 gradientOrientation(const Img& img, long x, long y) {
    const f32 result = std::atan2(img(x, y + 1) - img(x, y - 1), img(x + 1, y) - img(x - 1, y));
    return (f32)std::fmod((double)(result + 360.0f), 360.0);
}

u16 f32_to_u16_x86(f32 v) {
    // cvttss2si r32: out-of-range / NaN -> 0x80000000 ("integer indefinite"); then low 16 bits.
    int32_t i;
    if (v > -2147483904.0f && v < 2147483648.0f)
        i = (int32_t)v;
    else
        i = INT32_MIN;
    return (u16)((uint32_t)i & 0xffffu);
}

// algorithms.cpp:118-133 on 16x16 windows (x outer, y inner)
static void orientationHistogram36(const f32* ori, const f32* mag, const f32* gauss, long gstride,
                                   f32 (&bins)[36]) {
    for (int b = 0; b < 36; ++b) bins[b] = 0.0f;
    for (u16 x = 0; x < 16; ++x)
        for (u16 y = 0; y < 16; ++y) {
            const f32 sum = mag[x + y * 16] * gauss[x + y * gstride];
            u16 i = f32_to_u16_x86(std::floor(ori[x + y * 16] / 10.0f));
            i = i % 35;
            bins[i] += sum;
        }
}

// algorithms.cpp:210-223 (normalises by the SUM)
void normalizeVector(f32* v, int n) {
    f32 length = 0;
    for (int i = 0; i < n; ++i) length += v[i];
    if (length == 0) return;
    for (int i = 0; i < n; ++i) v[i] /= length;
}

// algorithms.cpp:153-178
f32 vertexParabola(u16 lnx, f32 lny, u16 px, f32 py, u16 rnx, f32 rny) {
    Mat a(3, 3);
    a(0, 0) = (f32)((double)lnx * (double)lnx);  // std::pow(u16, 2) in double, exact
    a(1, 0) = (f32)((double)px * (double)px);
    a(2, 0) = (f32)((double)rnx * (double)rnx);
    a(0, 1) = (f32)lnx;
    a(1, 1) = (f32)px;
    a(2, 1) = (f32)rnx;
    a(0, 2) = 0; a(1, 2) = 0; a(2, 2) = 0;
    Mat b(3, 1);
    b(0, 0) = lny; b(1, 0) = py; b(2, 0) = rny;
    Mat res(3, 1);
    linearSolve(a.v(), b.v(), res.v());
    return -res(1, 0) / (2.0f * res(0, 0));
}

// sift.cpp:295-345, body of the per-point loop.  true => p.filtered = true.
bool edgeResponseFiltered(const Img* const param[3], long x, long y) {
    f32 d[3], h[3][3];
    foDerivative(param, x, y, d);
    soDerivative(param, x, y, h);
    Mat neg(3, 3);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) neg(i, j) = h[i][j] * -1.0f;  // neg_sec_deriv *= -1
    Mat inverse_matrix(3, 3);
    if (!inverse(neg.v(), inverse_matrix.v())) return true;
    Mat deriv(3, 1);
    deriv(0, 0) = d[0]; deriv(1, 0) = d[1]; deriv(2, 0) = d[2];
    Mat extremum(3, 1);
    if (!linearSolve(inverse_matrix.v(), deriv.v(), extremum.v())) return true;
    if (extremum(0, 0) > 127.5f || extremum(1, 0) > 127.5f || extremum(2, 0) > 127.5f) return true;
    f32 func_val_extremum = dotRowCol(deriv.v().T(), extremum.v());
    func_val_extremum = (f32)((double)func_val_extremum * (0.5 + (double)(*param[1])(x, y)));
    if ((double)func_val_extremum < 7.65) return true;
    const f32 dxx = h[0][0], dyy = h[1][1];
    const f32 hessian_tr = dxx + dyy;
    const f32 prod = dxx * dyy;  // float product (mulss, bin@0x425508), then widened
    const f32 hessian_det = (f32)((double)prod - (double)h[0][1] * (double)h[0][1]);
    if (hessian_det < 0) return true;
    const f32 t = (f32)(std::pow(10 + 1, 2) / 10);
    if ((double)hessian_tr * (double)hessian_tr / (double)hessian_det > (double)t) return true;
    return false;
}

// ------------------------------------------------------------------------------------------
// sift::Sift (sift.cpp)
// ------------------------------------------------------------------------------------------

static void cleanup(std::vector<InterestPoint>& pts) {
    // sift.cpp:37-42 / 49-54: unstable std::sort, find_if, u16 size, resize
    std::sort(pts.begin(), pts.end(), InterestPoint::cmpByFilter);
    auto result = std::find_if(pts.begin(), pts.end(),
                               [](const InterestPoint& p) { return p.filtered; });
    const u16 size = (u16)std::distance(pts.begin(), result);
    pts.resize(size);
}

// sift.cpp:19-57
std::vector<InterestPoint> Sift::calculate(Img& img) {
    if (subpixel) img = increaseToNextLevel(img, 1.0f);
    _createDOGs(img);

    std::vector<InterestPoint> interestPoints;
    _findScaleSpaceExtrema(interestPoints);
    _eliminateEdgeResponses(interestPoints);
    trace.candidates = interestPoints;
    cleanup(interestPoints);
    trace.after_sort1 = interestPoints;

    _createGradientPyramids();
    _orientationAssignment(interestPoints);
    trace.after_orient = interestPoints;
    cleanup(interestPoints);
    trace.after_sort2 = interestPoints;

    _createDecriptors(interestPoints);
    return interestPoints;
}

// sift.cpp:381-417
void Sift::_createDOGs(Img& img) {
    if (!(_octaves > 0)) throw std::invalid_argument("Assertion `_octaves > 0' failed.");
    if (!(_dogsPerEpoch >= 3)) throw std::invalid_argument("Assertion `_dogsPerEpoch >= 3' failed.");
    _gaussians.assign((size_t)_octaves * (size_t)(_dogsPerEpoch + 1), OctaveElem());
    _dogs.assign((size_t)_octaves * (size_t)_dogsPerEpoch, OctaveElem());
    _magnitudes.assign(_gaussians.size(), Img());
    _orientations.assign(_gaussians.size(), Img());
    _weighting.assign(_gaussians.size(), Img());

    G(0, 0).scale = _sigma;
    G(0, 0).img = convolveWithGauss(img, _sigma);

    u16 exp = 0;
    for (i16 i = 0; i < (i16)_octaves; i++) {
        for (i16 j = 1; j < _dogsPerEpoch + 1; j++) {
            const f32 scale = (f32)(std::pow((double)_k, (double)exp) * (double)_sigma);
            G(i, j).scale = scale;
            G(i, j).img = convolveWithGauss(G(i, j - 1).img, scale);
            D(i, j - 1).scale = G(i, j).scale - G(i, j - 1).scale;
            D(i, j - 1).img = oracle::dog(G(i, j - 1).img, G(i, j).img);
            exp++;
        }
        if (i < (_octaves - 1)) {
            Img scaledElem =
                reduceToNextLevel(G(i, _dogsPerEpoch - 1).img, G(i, _dogsPerEpoch - 1).scale);
            G(i + 1, 0).scale = G(i, _dogsPerEpoch - 1).scale;
            G(i + 1, 0).img = std::move(scaledElem);
            exp -= 2;
        }
    }
}

// sift.cpp:348-379: 2x2x3 half-open neighbourhood, non-strict; order octave, dog, x outer, y inner
void Sift::_findScaleSpaceExtrema(std::vector<InterestPoint>& interestPoints) const {
    for (u16 e = 0; e < _octaves; e++) {
        for (u16 i = 1; i < _dogsPerEpoch - 1; i++) {
            const Img& cur = dog(e, i).img;
            const Img* three[3] = {&dog(e, i - 1).img, &cur, &dog(e, i + 1).img};
            for (i16 x = 1; x < cur.w - 1; x++) {
                for (i16 y = 1; y < cur.h - 1; y++) {
                    const f32 c = cur(x, y);
                    bool anyGreater = false, anySmaller = false;
                    for (int s = 0; s < 3; ++s)
                        for (int xx = x - 1; xx <= x; ++xx)
                            for (int yy = y - 1; yy <= y; ++yy) {
                                const f32 v = (*three[s])(xx, yy);
                                if (v > c) anyGreater = true;
                                if (v < c) anySmaller = true;
                            }
                    if (!anyGreater || !anySmaller) {
                        InterestPoint p;
                        p.x = (u16)x; p.y = (u16)y;
                        p.scale = dog(e, i).scale;
                        p.octave = e; p.index = i;
                        p.cand_id = (int32_t)interestPoints.size();
                        interestPoints.push_back(p);
                    }
                }
            }
        }
    }
}

// sift.cpp:288-346
void Sift::_eliminateEdgeResponses(std::vector<InterestPoint>& interestPoints) const {
    for (InterestPoint& p : interestPoints) {
        if (_faithful) {
            // sift.cpp:297-298 binds a const-ref to a temporary std::array => 3 deep image copies
            const Img c0 = dog(p.octave, p.index - 1).img, c1 = dog(p.octave, p.index).img,
                      c2 = dog(p.octave, p.index + 1).img;
            const Img* param[3] = {&c0, &c1, &c2};
            if (edgeResponseFiltered(param, p.x, p.y)) p.filtered = true;
        } else {
            const Img* param[3] = {&dog(p.octave, p.index - 1).img, &dog(p.octave, p.index).img,
                                   &dog(p.octave, p.index + 1).img};
            if (edgeResponseFiltered(param, p.x, p.y)) p.filtered = true;
        }
    }
}

// sift.cpp:130-160 for one level: interior pixels only, border stays 0
void Sift::_ensureGradient(int o, int i) {
    Img& mag = _magnitudes[(size_t)(o * levels() + i)];
    if (mag.w) return;
    const Img& g = gaussian(o, i).img;
    Img& ori = _orientations[(size_t)(o * levels() + i)];
    mag = Img(g.w, g.h);
    ori = Img(g.w, g.h);
    for (long x = 1; x < g.w - 1; ++x)
        for (long y = 1; y < g.h - 1; ++y) {
            mag(x, y) = gradientMagnitude(g, x, y);
            ori(x, y) = gradientOrientation(g, x, y);
        }
}

void Sift::_createGradientPyramids() {
    if (!_faithful) return;  // lean: levels are built on first use
    for (int o = 0; o < _octaves; ++o)
        for (int i = 0; i < levels(); ++i) _ensureGradient(o, i);
}

// sift.cpp:205-218: strict <, init 100, first minimum wins
void Sift::_findNearestGaussian(f32 scale, int& no, int& ni) const {
    f32 lowest_diff = 100;
    no = 0; ni = 0;
    for (u16 o = 0; o < _octaves; o++)
        for (u16 i = 0; i < levels(); i++) {
            const f32 cur_scale = std::abs(gaussian(o, i).scale - scale);
            if (cur_scale < lowest_diff) {
                lowest_diff = cur_scale;
                no = o; ni = i;
            }
        }
}

// sift.cpp:220-286
std::set<f32> Sift::_findPeaks(const f32 (&histo)[36]) const {
    std::set<f32> result;
    f32 peaks_only[36];
    std::memcpy(peaks_only, histo, sizeof(peaks_only));
    const u16 max_index = (u16)(std::max_element(peaks_only, peaks_only + 36) - peaks_only);
    const f32 range = (f32)((double)histo[max_index] * 0.8);
    for (f32& elem : peaks_only)
        if (elem < range) elem = -1;
    for (u16 i = 1; i < 36 - 1; i++)
        if (peaks_only[i] < peaks_only[i - 1] || peaks_only[i] < peaks_only[i + 1])
            peaks_only[i] = -1;

    auto vertexAt = [&](u16 i) {
        u16 lnx, rnx; f32 lny, rny;
        const u16 px = (u16)(i * 10 + 5);
        const f32 py = histo[i];
        if (i == 0) { lnx = (36 - 1) * 10 + 5; lny = histo[36 - 1]; }
        else        { lnx = (u16)((i - 1) * 10 + 5); lny = histo[i - 1]; }
        if (i == 36 - 1) { rnx = 5; rny = histo[0]; }
        else             { rnx = (u16)((i + 1) * 10 + 5); rny = histo[i + 1]; }
        return vertexParabola(lnx, lny, px, py, rnx, rny);
    };
    result.emplace(vertexAt(max_index));
    for (u16 i = 0; i < 36; i++)
        if (peaks_only[i] > -1 && i != max_index) result.emplace(vertexAt(i));
    return result;
}

// sift.cpp:163-203
void Sift::_orientationAssignment(std::vector<InterestPoint>& interestPoints) {
    const long region = 8;
    std::vector<InterestPoint> additional;
    for (InterestPoint& p : interestPoints) {
        int co, ci;
        _findNearestGaussian(p.scale, co, ci);
        const Img& closest = gaussian(co, ci).img;
        if ((p.x < region || p.x >= closest.w - region) ||
            (p.y < region || p.y >= closest.h - region)) {
            p.filtered = true;
            continue;
        }
        const long x0 = p.x - region, y0 = p.y - region;
        // sift.cpp:184 — dead blur of the 16x16 window with sigma = 1.5*scale; its only effect is
        // that it can throw (kernel longer than line) for large scales.
        {
            const f32 s = (f32)(1.5 * (double)p.scale);
            if (_faithful) {
                Img region_copy(16, 16);
                for (long y = 0; y < 16; ++y)
                    for (long x = 0; x < 16; ++x) region_copy(x, y) = closest(x0 + x, y0 + y);
                (void)convolveWithGauss(region_copy, s);
            } else {
                const Kernel1D ker = initGaussian((double)s);
                if (!(16 >= ker.radius + 1))
                    throw PreconditionViolation("separableConvolveX(): kernel longer than line\n");
            }
        }
        _ensureGradient(co, ci);
        const Img& oriL = *orientation(co, ci);
        const Img& magL = *magnitude(co, ci);
        f32 ori[256], mag[256];  // deep copies (MultiArray constructed from a view)
        for (long y = 0; y < 16; ++y)
            for (long x = 0; x < 16; ++x) {
                ori[x + y * 16] = oriL(x0 + x, y0 + y);
                mag[x + y * 16] = magL(x0 + x, y0 + y);
            }
        f32 histogram[36];
        orientationHistogram36(ori, mag, &closest(x0, y0), closest.w, histogram);
        const std::set<f32> peaks = _findPeaks(histogram);
        p.orientation = *(peaks.begin());
        if (peaks.size() > 1) {
            // sift.cpp:195: `peaks.begin()++` post-increments a temporary => starts at begin()
            for (auto iter = peaks.begin(); iter != peaks.end(); iter++) {
                InterestPoint temp = p;
                temp.orientation = *iter;
                additional.emplace_back(temp);
            }
        }
    }
    interestPoints.insert(interestPoints.end(), additional.begin(), additional.end());
}

// sift.cpp:60-110 (+ :113-128 whose result is discarded; only its in-place L1 normalise acts)
void Sift::_createDecriptors(std::vector<InterestPoint>& interestPoints) {
    const long region = 8;
    for (InterestPoint& p : interestPoints) {
        int co, ci;
        _findNearestGaussian(p.scale, co, ci);
        const Img& current = gaussian(co, ci).img;
        if (p.x < region || p.x > current.w - region || p.y < region || p.y > current.h - region) {
            p.filtered = true;
            continue;
        }
        _ensureGradient(co, ci);
        const long x0 = p.x - region, y0 = p.y - region;
        Img& oriL = _orientations[(size_t)(co * levels() + ci)];  // views => in-place mutation
        Img& magL = _magnitudes[(size_t)(co * levels() + ci)];

        for (u16 x = 0; x < 16; x++)
            for (u16 y = 0; y < 16; y++) oriL(x0 + x, y0 + y) += p.orientation;

        const Img* weighting;
        Img faithful_weighting;
        if (_faithful) {
            faithful_weighting = convolveWithGauss(current, 1.6f);
            weighting = &faithful_weighting;
        } else {
            Img& cache = _weighting[(size_t)(co * levels() + ci)];
            if (!cache.w) cache = convolveWithGauss(current, 1.6f);
            weighting = &cache;
        }
        for (u16 x = 0; x < 16; x++)
            for (u16 y = 0; y < 16; y++) magL(x0 + x, y0 + y) += (*weighting)(x, y);

        std::vector<f32> descriptors;
        descriptors.reserve(128);
        for (u16 x = 0; x < 16; x += 4)
            for (u16 y = 0; y < 16; y += 4) {
                f32 bins[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                // algorithms.cpp:135-150 on the 4x4 cell
                for (u16 cx = 0; cx < 4; cx++)
                    for (u16 cy = 0; cy < 4; cy++) {
                        const long X = x0 + x + cx, Y = y0 + y + cy;
                        const f32 sum = magL(X, Y) * current(X, Y);
                        u16 i = f32_to_u16_x86(std::floor(oriL(X, Y) / 45.0f));
                        i = i % 7;
                        bins[i] += sum;
                    }
                normalizeVector(bins, 8);  // sift.cpp:114; thresholded copy (:115-127) discarded
                descriptors.insert(descriptors.end(), bins, bins + 8);
            }
        p.descriptors = descriptors;
    }
}

}  // namespace oracle
