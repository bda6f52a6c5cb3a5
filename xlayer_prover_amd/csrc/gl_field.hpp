// Goldilocks field (p = 2^64 - 2^32 + 1) device arithmetic for gfx950.
//
// Replaces the arithmetic of the reference's (absent) goldilocks submodule as
// it is used on the Starks::genProof path (e.g. polinomial.hpp:140-229,
// friProve.cpp:192-217).  Written for CDNA4's 32-bit VALU: a field element is
// one u64 in a VGPR pair; sums are 32-bit carry chains (v_add_co/v_addc_co via
// __builtin_addc, which hipcc lowers without 64-bit compares), products are
// four v_mad_u64_u32 and the reduction uses 2^64 = 2^32 - 1, 2^96 = -1 (mod p).
// Measured on MI355X (profiles/r1/ubench_*.txt): add/sub ~6 VALU slots,
// mul ~35, mul by a constant power of two ~12-18.
//
// Representation:
//   "canon"  value in [0, p)
//   "any"    value in [0, 2^64), congruent to the element
// Stated per function; kernels store canonical values to HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gl {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 P = 0xFFFFFFFF00000001ULL;
constexpr u64 EPS = 0xFFFFFFFFULL;  // 2^64 mod p = 2^32 - 1

__device__ __forceinline__ u64 pack(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// hipcc (ROCm 7.2, gfx950) mis-combines two adjacent carry chains when the
// second operand of a link is the constant 0: (hi - 0 - borrow) + 0 + carry is
// folded into ONE v_addc whose carry-out is not the carry of the second add
// (caught by tests/test_gpu_parity.py::test_field_ops_exhaustive_edges).
// An empty asm makes the link's result opaque, which keeps the chains apart;
// it emits no instruction.
__device__ __forceinline__ void opaque(u32& x) { asm("" : "+v"(x)); }

// any -> canon : a >= p  <=>  a + EPS carries out of 64 bits (p + EPS = 2^64)
__device__ __forceinline__ u64 canon(u64 a)
{
    u32 c0, c1;
    u32 tlo = __builtin_addc((u32)a, 0xFFFFFFFFu, 0u, &c0);
    u32 thi = __builtin_addc((u32)(a >> 32), 0u, c0, &c1);
    opaque(tlo); opaque(thi);
    return c1 ? pack(tlo, thi) : a;
}

// canon + canon -> canon   (6 VALU: 2 add, 2 add, 2 select)
__device__ __forceinline__ u64 addc(u64 a, u64 b)
{
    u32 c1, c2, c3, c4;
    u32 lo = __builtin_addc((u32)a, (u32)b, 0u, &c1);
    u32 hi = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c1, &c2);
    opaque(lo); opaque(hi);
    u32 tlo = __builtin_addc(lo, 0xFFFFFFFFu, 0u, &c3);   // s - p  ==  s + EPS (mod 2^64)
    u32 thi = __builtin_addc(hi, 0u, c3, &c4);
    opaque(tlo); opaque(thi);
    return (c2 | c4) ? pack(tlo, thi) : pack(lo, hi);
}
// canon - canon -> canon   (5 VALU)
__device__ __forceinline__ u64 subc(u64 a, u64 b)
{
    u32 b1, b2, b3, b4;
    u32 lo = __builtin_subc((u32)a, (u32)b, 0u, &b1);
    u32 hi = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), b1, &b2);
    opaque(lo); opaque(hi);
    u32 m = b2 ? 0xFFFFFFFFu : 0u;                          // borrow: + p  ==  - EPS (mod 2^64)
    u32 rlo = __builtin_subc(lo, m, 0u, &b3);
    u32 rhi = __builtin_subc(hi, 0u, b3, &b4);
    opaque(rlo); opaque(rhi);
    return pack(rlo, rhi);
}
// any + canon -> any
__device__ __forceinline__ u64 add(u64 a, u64 b)
{
    u32 c1, c2, c3, c4;
    u32 lo = __builtin_addc((u32)a, (u32)b, 0u, &c1);
    u32 hi = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c1, &c2);
    opaque(lo); opaque(hi);
    u32 m = c2 ? 0xFFFFFFFFu : 0u;                          // carry: 2^64 = EPS (no second carry: b < p)
    u32 rlo = __builtin_addc(lo, m, 0u, &c3);
    u32 rhi = __builtin_addc(hi, 0u, c3, &c4);
    opaque(rlo); opaque(rhi);
    return pack(rlo, rhi);
}
// any - canon -> any
__device__ __forceinline__ u64 sub(u64 a, u64 b) { return subc(a, b); }  // same sequence, same argument

__device__ __forceinline__ u64 neg(u64 a) { return a ? P - a : 0; }  // canon -> canon

// lo + hl*EPS - hh  (mod p) -> any, where (hh:hl) = hi.   This is (hi:lo) mod p.
// Both wrap-arounds are settled together: s = lo + hl*EPS may carry (c: the true value is s + 2^64), l = s - hh may borrow (b: the true
// value is l - 2^64); net (c - b) * 2^64 = (c - b) * EPS.  c = 1, b = 0: s < 2^64 - 2^33 + 1 and l <= s, adding EPS cannot carry.  b = 1,
// c = 0: l >= 2^64 - 2^32 + 1, subtracting EPS cannot borrow.  b = c: nothing to do.
// Written in assembly: 8 VALU + 2 SALU (hipcc builds 11 VALU from the C form: it materialises the borrow with a v_cndmask instead of
// v_subbrev, compares instead of taking the multiply-add's carry-out, and spends four selects on the correction word).  Every field
// multiplication ends here.  HAZARD-SAFE by construction (round 3): on gfx940+ a VALU that reads an SGPR / VCC written by a VALU needs two
// wait states in between (LLVM's GCNHazardRecognizer, "VALUWriteSGPRVALURead", which cannot see inside an asm blob) -- every such producer /
// consumer pair below is two instructions apart: v_mad (c) .. v_sub_co, s_nop .. v_cndmask (c);  v_sub_co (b) .. s_nop, v_cndmask ..
// v_subbrev (b);  v_add_co (vcc) .. v_cndmask, s_nop .. v_addc_co (vcc).  The SALU reads of b and c and the VALU reads of the SALU-written
// x, y need none.  The result leaves as two 32-bit halves (what the next multiplication reads anyway): a 64-bit v_lshl_add_u64 for the
// correction was one instruction shorter on paper and 2 v_mov longer in the kernels (aligned register pairs), 14.6k -> 15.6k VALU per
// Poseidon permutation (profiles/r3/counters_lshl_add_variant.json).
__device__ __forceinline__ u64 reduce128(u64 lo, u64 hi)
{
    const u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 s, c;                                // c: lane mask: lo + hl * EPS carried out of 64 bits (the multiply-add's own carry-out)
    asm("v_mad_u64_u32 %[s], %[c], %[hl], -1, %[lo]" : [s] "=&v"(s), [c] "=&s"(c) : [hl] "v"(hl), [lo] "v"(lo));
    u32 l0, l1, t, r0, r1;
    u64 b, x, y;                             // b: lane mask: s - hh borrowed
    // + (c - b) * EPS as a two's complement word (t, r1): +EPS = (FFFFFFFF, 0), -EPS = (00000001, FFFFFFFF)
    asm("v_sub_co_u32 %[l0], %[b], %[s0], %[hh]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32 %[t], 1, -1, %[c]\n\t"
        "v_subbrev_co_u32 %[l1], %[b], 0, %[s1], %[b]\n\t"
        "s_xor_b64 %[x], %[b], %[c]\n\t"
        "s_andn2_b64 %[y], %[b], %[c]\n\t"
        "v_cndmask_b32 %[t], 0, %[t], %[x]\n\t"
        "v_add_co_u32 %[r0], vcc, %[l0], %[t]\n\t"
        "v_cndmask_b32 %[r1], 0, -1, %[y]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 %[r1], vcc, %[l1], %[r1], vcc"
        : [b] "=&s"(b), [l0] "=&v"(l0), [l1] "=&v"(l1), [t] "=&v"(t), [r0] "=&v"(r0), [r1] "=&v"(r1), [x] "=&s"(x), [y] "=&s"(y)
        : [c] "s"(c), [hh] "v"(hh), [s0] "v"((u32)s), [s1] "v"((u32)(s >> 32)) : "vcc", "scc");
    return pack(r0, r1);
}

// any * any -> any
__device__ __forceinline__ u64 mul(u64 a, u64 b)
{
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p0 = (u64)a0 * b0;
    const u64 p1 = (u64)a0 * b1 + (p0 >> 32);
    const u64 p2 = (u64)a1 * b0 + (u32)p1;
    const u64 hi = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32);
    const u64 lo = (p2 << 32) | (u32)p0;
    return reduce128(lo, hi);
}
__device__ __forceinline__ u64 sqr(u64 a) { return mul(a, a); }
__device__ __forceinline__ u64 mulc(u64 a, u64 b) { return canon(mul(a, b)); }

// canon(v) for a 65-bit value v = carry * 2^64 + r known to be < 2p:  (carry || r >= p) ? r + EPS : r
__device__ __forceinline__ u64 canon65(u64 r, bool carry)
{
    u32 c0, c1;
    u32 tlo = __builtin_addc((u32)r, 0xFFFFFFFFu, 0u, &c0);
    u32 thi = __builtin_addc((u32)(r >> 32), 0u, c0, &c1);
    opaque(tlo); opaque(thi);
    return (carry || c1) ? pack(tlo, thi) : r;
}

// any * 2^K (compile-time 0 <= K < 96) -> CANONICAL.   K = 32q + r, Y = a << r = (y2:y1:y0) (96 bits):
//   a * 2^K = Y * 2^(32q), reduced with 2^64 = EPS, 2^96 = -1, 2^128 = -2^32.
// Written with 64-bit shifts and v_mad_u64_u32 (one instruction each on gfx950, all half rate like
// the 32-bit carry ops they replace).
template <int K>
__device__ __forceinline__ u64 mul_pow2c(u64 a)
{
    static_assert(K >= 0 && K < 96, "shift out of range");
    if constexpr (K == 0) return canon(a);
    constexpr int q = K / 32, r = K % 32;
    if constexpr (q == 0) {
        // (y1:y0) + y2 * EPS   < 2^64 + 2^(32+r) < 2p
        const u64 z = a << r;
        const u32 y2 = (u32)(a >> (64 - r));
        const u64 s = z + (u64)y2 * 0xFFFFFFFFu;
        return canon65(s, s < z);
    } else if constexpr (q == 2) {
        // y0 * EPS - (y2:y1) ; both canonical (y0*EPS <= 2^64 - 2^33 + 1, (y2:y1) < 2^63)
        const u32 y0 = (u32)a << r;                    // r == 0: y0 = a_lo
        const u64 w = r ? a >> (32 - r) : a >> 32;     // (y2:y1)
        const u64 t = (u64)y0 * 0xFFFFFFFFu;
        return subc(t, w);
    } else {
        // (y0:0) + y1 * EPS - y2 = reduce128(lo = y0 << 32, hi = (y2:y1))
        const u32 y0 = (u32)a << r;
        const u64 w = r ? a >> (32 - r) : a >> 32;
        return canon(reduce128((u64)y0 << 32, w));
    }
}
// run-time shift (used off the hot path only)
__device__ __forceinline__ u64 mul_pow2(u64 a, unsigned k)
{
    if (k == 0) return a;
    if (k < 64) return reduce128(a << k, a >> (64 - k));
    return mul_pow2(reduce128(0, a), k - 64);
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

// ---------------------------------------------------------------- sums of (uniform Fp^3 coefficient) x (base-field value)
// What a step program's constraint accumulation is once its Horner chain is a dot product (tools/expr_opt.py): acc += k * v with k a
// challenge power (the same in every lane) and v the row's value of a constraint.  PRODUCTS WITHOUT CARRIES: the coefficient word is cut
// into limbs of 22 + 21 + 21 bits (scalar ALU: it is wave-uniform), the value into its 32-bit halves, and each of the six partial
// products (< 2^54) has an accumulator of its own: one v_mad_u64_u32 apiece -- 18 vector instructions for the Fp^3 x Fp multiply-add
// that three mulc + three addc spend ~63 on.  An accumulator takes 1 024 products before it can overflow (the code generator counts);
// fold() brings the sum back to three canonical words and leaves the accumulators as they are.
struct lazy3 {
    u64 a[3][6];                                      // per component: weights 2^0, 2^22, 2^43, 2^32, 2^54, 2^75
    u64 b[3];                                         // canonical words added at fold()
    __device__ __forceinline__ void init(u64 b0, u64 b1, u64 b2)
    {
        b[0] = b0; b[1] = b1; b[2] = b2;
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int i = 0; i < 6; i++) a[c][i] = 0;
    }
    __device__ __forceinline__ void mac1(int c, u64 k, u32 v0, u32 v1)
    {
        const u32 l0 = (u32)k & 0x3FFFFFu, l1 = (u32)(k >> 22) & 0x1FFFFFu, l2 = (u32)(k >> 43);
        a[c][0] += (u64)v0 * l0; a[c][1] += (u64)v0 * l1; a[c][2] += (u64)v0 * l2;
        a[c][3] += (u64)v1 * l0; a[c][4] += (u64)v1 * l1; a[c][5] += (u64)v1 * l2;
    }
    // += (k0, k1, k2) * v: any 64-bit words
    __device__ __forceinline__ void mac(u64 k0, u64 k1, u64 k2, u64 v)
    {
        const u32 v0 = (u32)v, v1 = (u32)(v >> 32);
        mac1(0, k0, v0, v1); mac1(1, k1, v0, v1); mac1(2, k2, v0, v1);
    }
    // += (k0, k1, k2) * (x0, x1, x2) in Fp[X] / (X^3 - X - 1), k any words, x CANONICAL: the nine products regrouped per component,
    // c0 = k0 x0 + k1 x2 + k2 x1, c1 = k0 x1 + k1 (x0 + x2) + k2 (x1 + x2), c2 = k0 x2 + k1 x1 + k2 (x0 + x2): three products per
    // accumulator (the code generator counts them as three), 54 multiply-adds and two field additions where mul3 + add3 spend ~180
    __device__ __forceinline__ void mac3(u64 k0, u64 k1, u64 k2, u64 x0, u64 x1, u64 x2)
    {
        const u64 x02 = addc(x0, x2), x12 = addc(x1, x2);
        mac1(0, k0, (u32)x0, (u32)(x0 >> 32)); mac1(0, k1, (u32)x2, (u32)(x2 >> 32)); mac1(0, k2, (u32)x1, (u32)(x1 >> 32));
        mac1(1, k0, (u32)x1, (u32)(x1 >> 32)); mac1(1, k1, (u32)x02, (u32)(x02 >> 32)); mac1(1, k2, (u32)x12, (u32)(x12 >> 32));
        mac1(2, k0, (u32)x2, (u32)(x2 >> 32)); mac1(2, k1, (u32)x1, (u32)(x1 >> 32)); mac1(2, k2, (u32)x02, (u32)(x02 >> 32));
    }
    // sum of a[c][i] 2^w[i] = X + Y 2^32, X = a0 + a1 2^22 + a2 2^43 < 2^108, Y likewise; Y is reduced first so that X + y 2^32 < 2^109
    __device__ __forceinline__ u64 fold1(int c) const
    {
        typedef unsigned __int128 u128;
        const u128 X = (u128)a[c][0] + ((u128)a[c][1] << 22) + ((u128)a[c][2] << 43);
        const u128 Y = (u128)a[c][3] + ((u128)a[c][4] << 22) + ((u128)a[c][5] << 43);
        const u64 y = reduce128((u64)Y, (u64)(Y >> 64));
        const u128 Z = X + ((u128)y << 32);
        return addc(canon(reduce128((u64)Z, (u64)(Z >> 64))), b[c]);
    }
    __device__ __forceinline__ void fold(u64& r0, u64& r1, u64& r2) const { r0 = fold1(0); r1 = fold1(1); r2 = fold1(2); }
};
// -(a mod p) for a wave-uniform word, in plain C so that it stays on the scalar ALU (canon() / neg() pin their operands to vector registers)
__device__ __forceinline__ u64 neg_any(u64 a)
{
    const u64 c = a >= P ? a - P : a;
    return c ? P - c : 0;
}

}  // namespace gl
