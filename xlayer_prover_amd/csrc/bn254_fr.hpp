// BN254 scalar-field (Fr) element arithmetic for the kernels of the fflonk prover (csrc/fflonk.hip): 8 x 32-bit limbs in Montgomery form
// (R = 2^256), the in-memory form of the reference's RawFr::Element (src/ffiasm/fr.hpp) viewed as 32-bit words.  The product is
// mont256.hpp's (product scanning, one accumulator: these kernels run at full occupancy).
#pragma once
#include "mont256.hpp"

namespace xlp {
namespace ffr {

typedef uint32_t u32;
typedef uint64_t u64;

struct Fr { u32 v[8]; };

struct FrParams {
    static constexpr u32 N[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr u32 N0INV = 0xefffffffu;
};
// R mod r (= one) and R^2 mod r
__device__ __constant__ u32 FR_ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
__device__ __constant__ u32 FR_R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};

__device__ __forceinline__ Fr zero() { Fr z; for (int i = 0; i < 8; i++) z.v[i] = 0; return z; }
__device__ __forceinline__ Fr one() { Fr z; for (int i = 0; i < 8; i++) z.v[i] = FR_ONE[i]; return z; }
__device__ __forceinline__ bool is_zero(const Fr& a) { u32 o = 0; for (int i = 0; i < 8; i++) o |= a.v[i]; return o == 0; }

__device__ __forceinline__ Fr add(const Fr& a, const Fr& b)
{
    u32 s[8], d[8];
    u32 c = 0, borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 t = (u64)a.v[i] + b.v[i] + c;
        s[i] = (u32)t;
        c = (u32)(t >> 32);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 t = (u64)s[i] - FrParams::N[i] - borrow;
        d[i] = (u32)t;
        borrow = (u32)(t >> 63);
    }
    const bool ge = c != 0 || borrow == 0;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = ge ? d[i] : s[i];
    return r;
}
__device__ __forceinline__ Fr sub(const Fr& a, const Fr& b)
{
    u32 d[8];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 s = (u64)a.v[i] - b.v[i] - borrow;
        d[i] = (u32)s;
        borrow = (u32)(s >> 63);
    }
    u32 c = 0;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 t = (u64)d[i] + (borrow ? FrParams::N[i] : 0u) + c;
        r.v[i] = (u32)t;
        c = (u32)(t >> 32);
    }
    return r;
}
__device__ __forceinline__ Fr neg(const Fr& a) { return is_zero(a) ? a : sub(zero(), a); }
__device__ __forceinline__ Fr mul(const Fr& a, const Fr& b)
{
    Fr r;
    mont::mul<FrParams, false>(r.v, a.v, b.v);
    return r;
}
__device__ __forceinline__ Fr sqr(const Fr& a) { return mul(a, a); }
__device__ __forceinline__ Fr to_mont(const Fr& raw)
{
    Fr r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = FR_R2[i];
    return mul(raw, r2);
}
__device__ __forceinline__ Fr from_mont(const Fr& a)
{
    Fr o = zero();
    o.v[0] = 1;
    return mul(a, o);
}
__device__ __forceinline__ Fr gld(const Fr* p)
{
    const uint4* q = (const uint4*)p;
    const uint4 a = q[0], b = q[1];
    Fr e;
    e.v[0] = a.x; e.v[1] = a.y; e.v[2] = a.z; e.v[3] = a.w; e.v[4] = b.x; e.v[5] = b.y; e.v[6] = b.z; e.v[7] = b.w;
    return e;
}
__device__ __forceinline__ void gst(Fr* p, const Fr& e)
{
    uint4* q = (uint4*)p;
    q[0] = make_uint4(e.v[0], e.v[1], e.v[2], e.v[3]);
    q[1] = make_uint4(e.v[4], e.v[5], e.v[6], e.v[7]);
}
// w^j from wpow[k] = w^(2^k)
__device__ __forceinline__ Fr pow_from_table(u64 j, const Fr* __restrict__ wpow)
{
    Fr acc = one();
    for (u32 k = 0; (j >> k) != 0; k++)
        if ((j >> k) & 1) acc = mul(acc, gld(&wpow[k]));
    return acc;
}
// a^(r - 2): the inverse (0 -> 0)
__device__ __forceinline__ Fr inverse(const Fr& a)
{
    // r - 2, little-endian 32-bit words
    const u32 e[8] = {0xefffffffu, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    Fr r = one();
    for (int i = 253; i >= 0; i--) {
        r = sqr(r);
        if ((e[i >> 5] >> (i & 31)) & 1) r = mul(r, a);
    }
    return r;
}

}  // namespace ffr
}  // namespace xlp
