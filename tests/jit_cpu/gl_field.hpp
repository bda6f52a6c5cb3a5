// CPU stand-in for csrc/gl_field.hpp (tests only): same names, plain 128-bit arithmetic.
#pragma once
#include <cstdint>
namespace gl {
typedef uint64_t u64;
typedef uint32_t u32;
constexpr u64 P = 0xFFFFFFFF00000001ULL;
// STRICT about representations (csrc/gl_field.hpp: "canon" < p, "any" < 2^64): a function that takes canonical words traps on anything else,
// and the functions that return "any" words return the NON-canonical representative whenever one exists (r + p < 2^64), so that generated text
// which hands an "any" word to something that needs a canonical one fails here instead of passing by luck.
static inline u64 canon(u64 a) { return a >= P ? a - P : a; }
static inline u64 need(u64 a) { if (a >= P) __builtin_trap(); return a; }
static inline u64 loose(u64 r) { return r < 0xFFFFFFFFULL ? r + P : r; }                       // the other representative of r, when it fits
static inline u64 addc(u64 a, u64 b) { return (u64)(((unsigned __int128)need(a) + need(b)) % P); }
static inline u64 subc(u64 a, u64 b) { return (u64)(((unsigned __int128)need(a) + P - need(b)) % P); }
static inline u64 mulc(u64 a, u64 b) { return (u64)(((unsigned __int128)canon(a) * canon(b)) % P); }
static inline u64 mul(u64 a, u64 b) { return loose(mulc(a, b)); }                              // any * any -> any
static inline u64 add(u64 a, u64 b) { return loose((u64)(((unsigned __int128)canon(a) + need(b)) % P)); }       // any + canon -> any
static inline u64 sub(u64 a, u64 b) { return loose((u64)(((unsigned __int128)canon(a) + P - need(b)) % P)); }   // any - canon -> any
static inline u64 neg(u64 a) { need(a); return a ? P - a : 0; }
struct e3 { u64 c[3]; };
static inline e3 mul3(const e3& a, const e3& b)   // Fp[x]/(x^3 - x - 1), schoolbook form; canonical in and out
{
    const u64 a0 = need(a.c[0]), a1 = need(a.c[1]), a2 = need(a.c[2]), b0 = need(b.c[0]), b1 = need(b.c[1]), b2 = need(b.c[2]);
    e3 r;
    r.c[0] = addc(addc(mulc(a0, b0), mulc(a1, b2)), mulc(a2, b1));
    r.c[1] = addc(addc(addc(mulc(a0, b1), mulc(a1, b0)), addc(mulc(a1, b2), mulc(a2, b1))), mulc(a2, b2));
    r.c[2] = addc(addc(mulc(a0, b2), mulc(a2, b0)), addc(mulc(a1, b1), mulc(a2, b2)));
    return r;
}
// stand-in for gl::lazy3: the same limb arithmetic, every accumulator addition checked against 64-bit overflow
struct lazy3 {
    u64 a[3][6];
    u64 b[3];
    void init(u64 b0, u64 b1, u64 b2)
    {
        b[0] = need(b0); b[1] = need(b1); b[2] = need(b2);
        for (int c = 0; c < 3; c++) for (int i = 0; i < 6; i++) a[c][i] = 0;
    }
    static void acc(u64& s, u64 x) { if (__builtin_add_overflow(s, x, &s)) __builtin_trap(); }
    void mac1(int c, u64 k, u32 v0, u32 v1)
    {
        const u32 l0 = (u32)k & 0x3FFFFFu, l1 = (u32)(k >> 22) & 0x1FFFFFu, l2 = (u32)(k >> 43);
        acc(a[c][0], (u64)v0 * l0); acc(a[c][1], (u64)v0 * l1); acc(a[c][2], (u64)v0 * l2);
        acc(a[c][3], (u64)v1 * l0); acc(a[c][4], (u64)v1 * l1); acc(a[c][5], (u64)v1 * l2);
    }
    void mac(u64 k0, u64 k1, u64 k2, u64 v)
    {
        const u32 v0 = (u32)v, v1 = (u32)(v >> 32);
        mac1(0, k0, v0, v1); mac1(1, k1, v0, v1); mac1(2, k2, v0, v1);
    }
    void mac3(u64 k0, u64 k1, u64 k2, u64 x0, u64 x1, u64 x2)
    {
        if (x0 >= P || x1 >= P || x2 >= P) __builtin_trap();    // the generator only passes canonical words
        const u64 x02 = addc(x0, x2), x12 = addc(x1, x2);
        mac1(0, k0, (u32)x0, (u32)(x0 >> 32)); mac1(0, k1, (u32)x2, (u32)(x2 >> 32)); mac1(0, k2, (u32)x1, (u32)(x1 >> 32));
        mac1(1, k0, (u32)x1, (u32)(x1 >> 32)); mac1(1, k1, (u32)x02, (u32)(x02 >> 32)); mac1(1, k2, (u32)x12, (u32)(x12 >> 32));
        mac1(2, k0, (u32)x2, (u32)(x2 >> 32)); mac1(2, k1, (u32)x1, (u32)(x1 >> 32)); mac1(2, k2, (u32)x02, (u32)(x02 >> 32));
    }
    u64 fold1(int c) const
    {
        typedef unsigned __int128 u128;
        const int w[6] = {0, 22, 43, 32, 54, 75};
        u128 s = b[c] % P;
        for (int i = 0; i < 6; i++) {
            u128 t = a[c][i] % P;
            for (int j = 0; j < w[i]; j++) t = (t * 2) % P;
            s = (s + t) % P;
        }
        return (u64)s;
    }
    void fold(u64& r0, u64& r1, u64& r2) const { r0 = fold1(0); r1 = fold1(1); r2 = fold1(2); }
};
static inline u64 neg_any(u64 a) { return neg(canon(a)); }
}  // namespace gl
