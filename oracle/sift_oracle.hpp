// ORACLE — TEST INFRASTRUCTURE ONLY (see vigra_restate.hpp header).  Parity pinned against the reference's own
// prebuilt binary (oracle/refexec, tests/test_ref_pins.py).
//
// CPU restatement of the reference's hot path sift::Sift::calculate() (sift.cpp:19-57) and the
// sift::alg::* helpers it calls (algorithms.cpp), including every behavioural quirk listed in
// SURVEY.md Appendix B.  Each function cites the reference file:line it follows.
#pragma once
#include <cstdint>
#include <set>
#include <string>
#include <vector>

#include "vigra_restate.hpp"

namespace oracle {

using u16 = unsigned short;
using i16 = short;
using f32 = float;

// interestpoint.hpp:13-63 (field order kept)
struct InterestPoint {
    f32 scale = 0;
    u16 octave = 0;
    u16 index = 0;
    bool filtered = false;
    u16 x = 0, y = 0;       // Point<u16,u16> loc
    f32 orientation = 0;    // uninitialised by the reference ctor; 0 here for determinism
    std::vector<f32> descriptors;
    int32_t cand_id = -1;   // oracle-only bookkeeping: position in the extrema scan order
    static bool cmpByFilter(const InterestPoint& a, const InterestPoint& b) {
        return !a.filtered && b.filtered;  // interestpoint.hpp:57-62
    }
};

struct OctaveElem {  // octaveelem.hpp:12-25
    f32 scale = 0;
    Img img;
};

struct Params {
    u16 dogsPerEpoch = 3;
    u16 octaves = 3;
    f32 sigma = 1.6f;
    f32 k = 1.41421356237309504880f;  // (float)std::sqrt(2)
    bool subpixel = false;
};

// Intermediates captured for stage-by-stage parity tests.
struct Trace {
    std::vector<InterestPoint> candidates;    // after _findScaleSpaceExtrema + _eliminateEdgeResponses (flags set)
    std::vector<InterestPoint> after_sort1;   // after first cleanup (sift.cpp:37-42)
    std::vector<InterestPoint> after_orient;  // after _orientationAssignment (incl. appended extras)
    std::vector<InterestPoint> after_sort2;   // after second cleanup (sift.cpp:49-54)
};

class Sift {
public:
    const bool subpixel;
    // faithful_cost = true keeps the reference's redundant work (3 deep DoG copies per candidate,
    // sift.cpp:297-298; dead 16x16 blur, sift.cpp:184; full-level re-blur per keypoint,
    // sift.cpp:87; gradient maps for every level, sift.cpp:130-160) so it can be TIMED as "the
    // reference CPU path".  false hoists that work; results are identical.
    explicit Sift(const Params& p, bool faithful_cost = false)
        : subpixel(p.subpixel), _sigma(p.sigma), _k(p.k), _dogsPerEpoch(p.dogsPerEpoch),
          _octaves(p.octaves), _faithful(faithful_cost) {}

    std::vector<InterestPoint> calculate(Img& img);

    // pyramid accessors for tests
    int octaves() const { return _octaves; }
    int levels() const { return _dogsPerEpoch + 1; }
    const OctaveElem& gaussian(int o, int i) const { return _gaussians[(size_t)(o * levels() + i)]; }
    const OctaveElem& dog(int o, int i) const { return _dogs[(size_t)(o * _dogsPerEpoch + i)]; }
    const Img* magnitude(int o, int i) const {
        const auto& g = _magnitudes[(size_t)(o * levels() + i)];
        return g.w ? &g : nullptr;
    }
    const Img* orientation(int o, int i) const {
        const auto& g = _orientations[(size_t)(o * levels() + i)];
        return g.w ? &g : nullptr;
    }
    Trace trace;

private:
    const f32 _sigma, _k;
    const u16 _dogsPerEpoch, _octaves;
    const bool _faithful;
    std::vector<OctaveElem> _gaussians, _dogs;  // Matrix<OctaveElem>, index o*height+i (matrix.hpp:58)
    std::vector<Img> _magnitudes, _orientations;
    std::vector<Img> _weighting;  // lean-mode cache of convolveWithGauss(level, 1.6)

    OctaveElem& G(int o, int i) { return _gaussians[(size_t)(o * levels() + i)]; }
    OctaveElem& D(int o, int i) { return _dogs[(size_t)(o * _dogsPerEpoch + i)]; }

    void _createDOGs(Img& img);
    void _findScaleSpaceExtrema(std::vector<InterestPoint>& pts) const;
    void _eliminateEdgeResponses(std::vector<InterestPoint>& pts) const;
    void _createGradientPyramids();
    void _ensureGradient(int o, int i);
    void _orientationAssignment(std::vector<InterestPoint>& pts);
    void _findNearestGaussian(f32 scale, int& o, int& i) const;
    std::set<f32> _findPeaks(const f32 (&histo)[36]) const;
    void _createDecriptors(std::vector<InterestPoint>& pts);
};

// sift::alg free functions (algorithms.cpp), exposed for known-answer tests
Img reduceToNextLevel(const Img& img, f32 sigma);    // algorithms.cpp:24-36
Img increaseToNextLevel(const Img& img, f32 sigma);  // algorithms.cpp:38-49
Img dog(const Img& lower, const Img& higher);        // algorithms.cpp:52-64
void foDerivative(const Img* const img[3], long x, long y, f32 d[3]);      // algorithms.cpp:66-77
void soDerivative(const Img* const img[3], long x, long y, f32 h[3][3]);   // algorithms.cpp:79-106
f32 gradientMagnitude(const Img& img, long x, long y);     // algorithms.cpp:108-111
f32 gradientOrientation(const Img& img, long x, long y);   // algorithms.cpp:113-116
f32 vertexParabola(u16 lnx, f32 lny, u16 px, f32 py, u16 rnx, f32 rny);   // algorithms.cpp:153-178
void normalizeVector(f32* v, int n);                       // algorithms.cpp:210-223
// returns true if the candidate is FILTERED by sift.cpp:295-345's per-point body
bool edgeResponseFiltered(const Img* const dogs3[3], long x, long y);
// (u16) conversion of a float as the x86-64 reference binary performs it (cvttss2si + 16-bit
// truncation); defined for every input instead of C++'s UB for out-of-range values.
u16 f32_to_u16_x86(f32 v);

}  // namespace oracle
