// 256-bit Montgomery multiplication for gfx950, shared by the BN254 scalar-field (bn128.hip) and base-field (bn254_msm.hip) code.
//
// Finely integrated product scanning (FIPS): the 16 result columns are summed one after the other in a 96-bit accumulator
// (lo:mid as one 64-bit register pair + hi), every 32 x 32 product is ONE v_mad_u64_u32 whose carry-out (VCC) feeds ONE
// v_addc_co_u32 on the top word -- two VALU instructions per multiply-accumulate, 128 of them per product -- instead of
// the compare-and-add sequences hipcc builds for `u64 += (u64)a * b` chains (measured: 668 -> ~350 VALU per Montgomery product;
// these kernels sit at ~75 % of the VALU issue peak, so instructions are time).  The modulus limbs are compile-time constants
// and ride in SGPRs (one scalar operand per VOP3 instruction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xlp {
namespace mont {

typedef uint32_t u32;
typedef uint64_t u64;

// acc (lo64, hi32) += a * b.  Two wait states between the multiply-add that writes VCC and the add that reads it: gfx940+ does not
// interlock a VALU-written SGPR / VCC against the next VALU's read (LLVM GCNHazardRecognizer, "VALUWriteSGPRVALURead": 2 wait states),
// and the compiler cannot see inside an asm blob.
__device__ __forceinline__ void mac(u64& lo, u32& hi, u32 a, u32 b)
{
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
}
// acc += a * K, K a compile-time constant (modulus limb) in an SGPR
__device__ __forceinline__ void mac_k(u64& lo, u32& hi, u32 a, u32 k)
{
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "s"(k) : "vcc");
}

// Two multiply-accumulates at once: acc1 += a1 * b1 (VGPR x VGPR), acc2 += a2 * k2 (VGPR x SGPR constant); acc1 and acc2 may be the SAME
// accumulator (then the second multiply-add simply continues the first).  The two carries live in different registers (VCC and an SGPR
// pair) and each is consumed two instructions after it is produced -- M1 M2 nop A1 A2 -- so the pair costs five issue slots instead of the
// six of two separate mac() calls with their own wait states.
__device__ __forceinline__ void mac2(u64& lo1, u32& hi1, u32 a1, u32 b1, u64& lo2, u32& hi2, u32 a2, u32 k2)
{
    u64 c2;
    asm("v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %2, %4, %7, %8, %2\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_addc_co_u32 %3, %4, 0, %3, %4"
        : "+v"(lo1), "+v"(hi1), "+v"(lo2), "+v"(hi2), "=&s"(c2) : "v"(a1), "v"(b1), "v"(a2), "s"(k2) : "vcc");
}
// the same on ONE accumulator: acc += a1 * b1 + a2 * k2
__device__ __forceinline__ void mac2s(u64& lo, u32& hi, u32 a1, u32 b1, u32 a2, u32 k2)
{
    u64 c2;
    asm("v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, %2, %5, %6, %0\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_addc_co_u32 %1, %2, 0, %1, %2"
        : "+v"(lo), "+v"(hi), "=&s"(c2) : "v"(a1), "v"(b1), "v"(a2), "s"(k2) : "vcc");
}

// acc += a1 * k1 + a2 * k2 with both multipliers wave-uniform (SGPRs: table entries, modulus limbs)
__device__ __forceinline__ void mac2kk(u64& lo, u32& hi, u32 a1, u32 k1, u32 a2, u32 k2)
{
    u64 c2;
    asm("v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, %2, %5, %6, %0\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_addc_co_u32 %1, %2, 0, %1, %2"
        : "+v"(lo), "+v"(hi), "=&s"(c2) : "v"(a1), "s"(k1), "v"(a2), "s"(k2) : "vcc");
}
// a run of n multiply-accumulates acc += a(i) * k(i), i = 0 .. n - 1 (n a compile-time constant after unrolling), two at a time
template <class FA, class FK>
__device__ __forceinline__ void mac_run_k(u64& lo, u32& hi, int n, FA a, FK k)
{
    int i = 0;
#pragma unroll
    for (; i + 1 < n; i += 2) mac2kk(lo, hi, a(i), k(i), a(i + 1), k(i + 1));
    if (i < n) mac_k(lo, hi, a(i), k(i));
}

// (lo, hi) += (lo2, hi2)
__device__ __forceinline__ void acc_add(u64& lo, u32& hi, u64 lo2, u32 hi2)
{
    const u64 s = lo + lo2;
    hi += hi2 + (s < lo2 ? 1u : 0u);
    lo = s;
}

// P: struct with static constexpr u32 N[8] (modulus, little-endian limbs) and N0INV (-N^-1 mod 2^32).
// r = a * b / 2^256 mod N;  b < N, a < 2^256;  r < N.
// SPLIT: two accumulators per column -- the a * b products and the m * N products -- are independent dependency chains (each
// multiply-accumulate waits for the previous one on the same accumulator).  Pays where a lone wave per SIMD has nothing else to
// issue meanwhile (the Poseidon-BN128 kernels: 1.24x -> 1.52x on the recursivef tree); costs 5 % where occupancy already hides the
// latency (FFT, MSM), which therefore use one accumulator.
template <class P, bool SPLIT = false>
__device__ __forceinline__ void mul(u32 (&r)[8], const u32 (&a)[8], const u32 (&b)[8])
{
    u32 m[8];
    u64 lo = 0;
    u32 hi = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 lo2 = 0;
        u32 hi2 = 0;
#pragma unroll
        for (int j = 0; j < i; j++) {
            if constexpr (SPLIT) mac2(lo, hi, a[j], b[i - j], lo2, hi2, m[j], P::N[i - j]);
            else mac2s(lo, hi, a[j], b[i - j], m[j], P::N[i - j]);
        }
        mac(lo, hi, a[i], b[0]);
        if (SPLIT && i > 0) acc_add(lo, hi, lo2, hi2);
        m[i] = (u32)lo * P::N0INV;
        mac_k(lo, hi, m[i], P::N[0]);
        lo = (lo >> 32) | ((u64)hi << 32);
        hi = 0;
    }
    u32 t[9];
#pragma unroll
    for (int i = 8; i < 16; i++) {
        u64 lo2 = 0;
        u32 hi2 = 0;
#pragma unroll
        for (int j = i - 7; j < 8; j++) {
            if constexpr (SPLIT) mac2(lo, hi, a[j], b[i - j], lo2, hi2, m[j], P::N[i - j]);
            else mac2s(lo, hi, a[j], b[i - j], m[j], P::N[i - j]);
        }
        if (SPLIT && i < 15) acc_add(lo, hi, lo2, hi2);
        t[i - 8] = (u32)lo;
        lo = (lo >> 32) | ((u64)hi << 32);
        hi = 0;
    }
    t[8] = (u32)lo;
    // t < 2N: one conditional subtraction
    u32 d[8];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 s = (u64)t[i] - P::N[i] - borrow;
        d[i] = (u32)s;
        borrow = (u32)(s >> 63);
    }
    const bool ge = t[8] != 0 || borrow == 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = ge ? d[i] : t[i];
}

// ---- sums of products with ONE Montgomery reduction (the dot products of the Poseidon-BN128 linear layers)
// acc (16 limbs) += k * b.   k: wave-uniform table element (SGPRs), b: a lane's value.  The caller bounds the sum below 2^512.
__device__ __forceinline__ void mac512(u32 (&acc)[16], const u32 (&k)[8], const u32 (&b)[8])
{
    u64 lo = 0;
    u32 hi = 0;
#pragma unroll
    for (int c = 0; c < 16; c++) {
        {
            // terms of column c: acc[c] * 1 and b[c - x] * k[x] for x0 <= x <= x1
            const int x0 = c > 7 ? c - 7 : 0, x1 = c < 7 ? c : 7;
            mac_run_k(lo, hi, x1 - x0 + 2, [&](int i) { return i == 0 ? acc[c] : b[c - (x0 + i - 1)]; }, [&](int i) { return i == 0 ? 1u : k[x0 + i - 1]; });
        }
        acc[c] = (u32)lo;
        lo = (lo >> 32) | ((u64)hi << 32);
        hi = 0;
    }
}
// r = t / 2^256 mod N for t (16 limbs) < SUBS * N * 2^256 - N * 2^256 + ... : the Montgomery reduction of a sum of products, followed by
// SUBS conditional subtractions (a sum of n products of values below N needs ceil((n * N / 2^256) + 1) - 1 of them; the result is canonical).
template <class P, int SUBS>
__device__ __forceinline__ void redc512(u32 (&r)[8], const u32 (&t)[16])
{
    u32 m[8];
    u64 lo = 0;
    u32 hi = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        mac_run_k(lo, hi, i + 1, [&](int q) { return q == 0 ? t[i] : m[q - 1]; }, [&](int q) { return q == 0 ? 1u : P::N[i - (q - 1)]; });
        m[i] = (u32)lo * P::N0INV;
        mac_k(lo, hi, m[i], P::N[0]);
        lo = (lo >> 32) | ((u64)hi << 32);
        hi = 0;
    }
    u32 v[9];
#pragma unroll
    for (int i = 8; i < 16; i++) {
        mac_run_k(lo, hi, 16 - i, [&](int q) { return q == 0 ? t[i] : m[i - 7 + q - 1]; }, [&](int q) { return q == 0 ? 1u : P::N[i - (i - 7 + q - 1)]; });
        v[i - 8] = (u32)lo;
        lo = (lo >> 32) | ((u64)hi << 32);
        hi = 0;
    }
    v[8] = (u32)lo;
#pragma unroll
    for (int s = 0; s < SUBS; s++) {
        u32 d[9];
        u32 borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u64 x = (u64)v[i] - P::N[i] - borrow;
            d[i] = (u32)x;
            borrow = (u32)(x >> 63);
        }
        const u64 x = (u64)v[8] - borrow;
        d[8] = (u32)x;
        const bool ge = (x >> 63) == 0;
#pragma unroll
        for (int i = 0; i < 9; i++) v[i] = ge ? d[i] : v[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = v[i];
}

}  // namespace mont
}  // namespace xlp
