// Scalar type names of the drop-in API.  The reference's public headers (sift.hpp, interestpoint.hpp, point.hpp)
// are written in terms of these eleven names (/root/reference/types.hpp:4-15), so code that includes them keeps
// compiling; note that the "32-bit" names are `long`, i.e. 64 bits wide on LP64, exactly as there.
#ifndef SIFT_AMD_TYPES_HPP
#define SIFT_AMD_TYPES_HPP

typedef float f32_t;
typedef double f64_t;
typedef long double f80_t;

typedef char i8_t;
typedef short i16_t;
typedef long i32_t;
typedef long long i64_t;

typedef unsigned char u8_t;
typedef unsigned short u16_t;
typedef unsigned long u32_t;
typedef unsigned long long u64_t;

static_assert(sizeof(u16_t) == 2 && sizeof(f32_t) == 4, "the C ABI of libsift_hip.so assumes these widths");

#endif  // SIFT_AMD_TYPES_HPP
