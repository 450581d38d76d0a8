// sift::Point<T, U> — two coordinates of possibly different types (/root/reference/point.hpp:11-27).
#ifndef SIFT_AMD_POINT_HPP
#define SIFT_AMD_POINT_HPP
#include "types.hpp"
namespace sift {
template <typename T, typename U>
class Point {
public:
    T x;
    U y;
    Point() = default;
    Point(T x_, U y_) : x(x_), y(y_) {}
};
}  // namespace sift
#endif
