// Minimal stand-in for vigra::MultiArray<2, float> at the drop-in boundary: contiguous float
// image, x fastest (img(x, y) = data[x + y*width]), the layout Sift::calculate() receives in the
// reference (/root/reference/sift.hpp:78).  With -DSIFT_WITH_VIGRA, sift.hpp adds an overload that
// takes the real vigra::MultiArray<2, f32_t>&.
#ifndef SIFT_AMD_IMAGE2F_HPP
#define SIFT_AMD_IMAGE2F_HPP
#include <cstddef>
#include <vector>
namespace sift {
class Image2f {
public:
    Image2f() = default;
    Image2f(std::ptrdiff_t width, std::ptrdiff_t height) : w_(width), h_(height), d_((size_t)width * (size_t)height, 0.0f) {}
    std::ptrdiff_t width() const { return w_; }
    std::ptrdiff_t height() const { return h_; }
    float& operator()(std::ptrdiff_t x, std::ptrdiff_t y) { return d_[(size_t)x + (size_t)y * (size_t)w_]; }
    const float& operator()(std::ptrdiff_t x, std::ptrdiff_t y) const { return d_[(size_t)x + (size_t)y * (size_t)w_]; }
    float* data() { return d_.data(); }
    const float* data() const { return d_.data(); }
    void reshape(std::ptrdiff_t width, std::ptrdiff_t height) { w_ = width; h_ = height; d_.assign((size_t)width * (size_t)height, 0.0f); }
private:
    std::ptrdiff_t w_ = 0, h_ = 0;
    std::vector<float> d_;
};
}  // namespace sift
#endif
