// Host-side pieces of the hot path that are not per-pixel work: Gaussian tap tables, resampling
// index maps, the scale schedule of Sift::_createDOGs and the two "cleanup" steps of
// Sift::calculate (/root/reference/sift.cpp:37-42, 49-54).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace sift_hip {

// Kernel1D<float>::initGaussian as Vigra 1.11 computes it (called from
// /root/reference/algorithms.cpp:13-14): float taps via expf, normalised by a float running sum,
// radius (int)(3*sigma + 0.5) (>= 1).  Returns false when sigma < 0 (Vigra precondition).
bool gauss_taps(float sigma, std::vector<float>& taps, int& radius);

// vigra::resizeLineNoInterpolation index rule (accumulated double; algorithms.cpp:33,46).
std::vector<int> resize_index_map(int wold, int wnew);

// Order in which libstdc++'s std::sort(first, last, InterestPoint::cmpByFilter)
// (/root/reference/sift.cpp:37, interestpoint.hpp:57-62) leaves n elements with the given
// filtered flags: perm[i] = original index of the element that ends at position i.
void sort_by_filter(const uint8_t* flags, int n, std::vector<uint32_t>& perm);

// sift.cpp:37-42: sort, find first filtered, `u16_t size`, resize.  Returns the surviving original
// indices in their post-sort order.
void cleanup_survivors(const uint8_t* flags, int n, std::vector<uint32_t>& survivors);

// Run fn(i) for i in [0, n) on up to `threads` host threads.
void parallel_for(int n, int threads, void (*fn)(int, void*), void* arg);

}  // namespace sift_hip
