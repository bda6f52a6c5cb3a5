// CPU stand-in for csrc/gl_field.hpp (tests only): same names, plain 128-bit arithmetic.
#pragma once
#include <cstdint>
namespace gl {
typedef uint64_t u64;
typedef uint32_t u32;
constexpr u64 P = 0xFFFFFFFF00000001ULL;
static inline u64 canon(u64 a) { return a >= P ? a - P : a; }
static inline u64 addc(u64 a, u64 b) { return (u64)(((unsigned __int128)canon(a) + canon(b)) % P); }
static inline u64 subc(u64 a, u64 b) { return (u64)(((unsigned __int128)canon(a) + P - canon(b)) % P); }
static inline u64 mulc(u64 a, u64 b) { return (u64)(((unsigned __int128)canon(a) * canon(b)) % P); }
static inline u64 neg(u64 a) { a = canon(a); return a ? P - a : 0; }
struct e3 { u64 c[3]; };
static inline e3 mul3(const e3& a, const e3& b)   // Fp[x]/(x^3 - x - 1), schoolbook form
{
    const u64 a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], b0 = b.c[0], b1 = b.c[1], b2 = b.c[2];
    e3 r;
    r.c[0] = addc(addc(mulc(a0, b0), mulc(a1, b2)), mulc(a2, b1));
    r.c[1] = addc(addc(addc(mulc(a0, b1), mulc(a1, b0)), addc(mulc(a1, b2), mulc(a2, b1))), mulc(a2, b2));
    r.c[2] = addc(addc(mulc(a0, b2), mulc(a2, b0)), addc(mulc(a1, b1), mulc(a2, b2)));
    return r;
}
}  // namespace gl
