// Goldilocks field (p = 2^64 - 2^32 + 1) device arithmetic for gfx950.
//
// Replaces the arithmetic of the reference's (absent) goldilocks submodule as
// it is used on the Starks::genProof path (e.g. polinomial.hpp:140-229,
// friProve.cpp:192-217).  Written for CDNA4's 32-bit VALU: a field element is
// one u64 in a VGPR pair; products are built from 32x32->64 multiply-adds
// (v_mad_u64_u32) and reduced with 2^64 = 2^32 - 1, 2^96 = -1 (mod p).
//
// Representation invariants (documented per function):
//   "canon"  value in [0, p)
//   "any"    value in [0, 2^64), congruent to the element (lazy form)
// Kernels canonicalise once, on the store to HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gl {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 P = 0xFFFFFFFF00000001ULL;
constexpr u64 EPS = 0xFFFFFFFFULL;  // 2^64 mod p = 2^32 - 1

__device__ __forceinline__ u64 canon(u64 a) { return a >= P ? a - P : a; }

// any + canon -> any   (at least one operand must be canonical)
__device__ __forceinline__ u64 add(u64 a, u64 b)
{
    u64 s = a + b;
    return s < a ? s + EPS : s;
}
// any - canon -> any
__device__ __forceinline__ u64 sub(u64 a, u64 b)
{
    u64 d = a - b;
    return a < b ? d - EPS : d;
}
// canon + canon -> canon
__device__ __forceinline__ u64 addc(u64 a, u64 b)
{
    u64 s = a + b;
    return (s < a || s >= P) ? s - P : s;
}
// canon - canon -> canon
__device__ __forceinline__ u64 subc(u64 a, u64 b)
{
    u64 d = a - b;
    return a < b ? d + P : d;
}
__device__ __forceinline__ u64 neg(u64 a)  // canon -> canon
{
    return a ? P - a : 0;
}

// (hi:lo) 128-bit -> any.   x = lo + hi_lo*2^64 + hi_hi*2^96 = lo + hi_lo*(2^32-1) - hi_hi
__device__ __forceinline__ u64 reduce128(u64 lo, u64 hi)
{
    u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= EPS;                 // borrow: +p  (no second borrow: lo-hh > -2^32)
    u64 t1 = ((u64)hl << 32) - hl;          // hl * (2^32 - 1)
    u64 r = t0 + t1;
    if (r < t1) r += EPS;                   // carry: 2^64 = 2^32-1 (no second carry)
    return r;
}

// any * any -> any
__device__ __forceinline__ u64 mul(u64 a, u64 b)
{
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p0 = (u64)a0 * b0;
    u64 p1 = (u64)a0 * b1 + (p0 >> 32);
    u64 p2 = (u64)a1 * b0 + (u32)p1;
    u64 hi = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32);
    u64 lo = (p2 << 32) | (u32)p0;
    return reduce128(lo, hi);
}
__device__ __forceinline__ u64 sqr(u64 a) { return mul(a, a); }
__device__ __forceinline__ u64 mulc(u64 a, u64 b) { return canon(mul(a, b)); }

// any * 2^k (0 <= k < 96, compile-time friendly) -> any
__device__ __forceinline__ u64 mul_pow2(u64 a, unsigned k)
{
    // a*2^k as (hi:lo) 128-bit when k < 64, else shift in two steps
    if (k == 0) return a;
    if (k < 64) {
        u64 lo = a << k, hi = a >> (64 - k);
        return reduce128(lo, hi);
    }
    // k in [64,96): a*2^64*2^(k-64); 2^64 = EPS
    u64 t = reduce128(0, a);                // a * 2^64
    return mul_pow2(t, k - 64);
}

__device__ __forceinline__ u64 pow(u64 a, u64 e)
{
    u64 r = 1;
    while (e) { if (e & 1) r = mul(r, a); a = mul(a, a); e >>= 1; }
    return r;
}

// ---------------------------------------------------------------- cubic ext
// Fp[x]/(x^3 - x - 1), Karatsuba form of polinomial.hpp:195-205; canon in/out.
struct e3 { u64 c[3]; };
__device__ __forceinline__ e3 mul3(const e3& a, const e3& b)
{
    u64 A = mulc(addc(a.c[0], a.c[1]), addc(b.c[0], b.c[1]));
    u64 B = mulc(addc(a.c[0], a.c[2]), addc(b.c[0], b.c[2]));
    u64 C = mulc(addc(a.c[1], a.c[2]), addc(b.c[1], b.c[2]));
    u64 D = mulc(a.c[0], b.c[0]);
    u64 E = mulc(a.c[1], b.c[1]);
    u64 F = mulc(a.c[2], b.c[2]);
    u64 G = subc(D, E);
    e3 r;
    r.c[0] = subc(addc(C, G), F);
    r.c[1] = subc(subc(subc(addc(A, C), E), E), D);
    r.c[2] = subc(B, G);
    return r;
}
__device__ __forceinline__ e3 add3(const e3& a, const e3& b)
{
    e3 r; for (int i = 0; i < 3; i++) r.c[i] = addc(a.c[i], b.c[i]); return r;
}
__device__ __forceinline__ e3 mul3s(const e3& a, u64 s)  // ext * base, s any
{
    e3 r; for (int i = 0; i < 3; i++) r.c[i] = mulc(a.c[i], s); return r;
}

}  // namespace gl
