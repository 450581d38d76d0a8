// Fixed-width aliases with the reference's names (/root/reference/types.hpp:4-15).
#ifndef SIFT_AMD_TYPES_HPP
#define SIFT_AMD_TYPES_HPP
using u8_t = unsigned char;
using i8_t = char;
using u16_t = unsigned short int;
using i16_t = short int;
using u32_t = unsigned long int;   // 64-bit on LP64, as in the reference
using i32_t = long int;
using u64_t = unsigned long long int;
using i64_t = long long int;
using f32_t = float;
using f64_t = double;
using f80_t = long double;
#endif
