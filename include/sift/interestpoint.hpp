// sift::InterestPoint with the reference's fields in the reference's order
// (/root/reference/interestpoint.hpp:13-63).
#ifndef SIFT_AMD_INTERESTPOINT_HPP
#define SIFT_AMD_INTERESTPOINT_HPP
#include <set>
#include <vector>

#include "point.hpp"
#include "types.hpp"
namespace sift {
class InterestPoint {
public:
    f32_t scale;
    u16_t octave;
    u16_t index;            // DoG index inside the octave
    bool filtered = false;
    Point<u16_t, u16_t> loc;
    f32_t orientation;
    std::vector<f32_t> descriptors;

    InterestPoint() = default;
    explicit InterestPoint(Point<u16_t, u16_t> loc_, f32_t scale_, u16_t octave_, u16_t index_)
        : scale(scale_), octave(octave_), index(index_), loc(loc_) {}

    // true iff a is kept and b is filtered
    static bool cmpByFilter(const InterestPoint& a, const InterestPoint& b) { return !a.filtered && b.filtered; }
};
}  // namespace sift
#endif
