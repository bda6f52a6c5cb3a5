// DFT16 / DFT8 over Goldilocks on the matrix pipe: what the MFMA kernels of extendPol share (ntt_lde_mid_mfma.hip, ntt_lde_top_mfma.hip).
// See the header of ntt_lde_mid_mfma.hip for the idea; tools/ubench_dft16_mfma.hip for the measurement that preceded it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gl_field.hpp"
#include "xlp_internal.hpp"

namespace xlp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// 1 pad row per 16 and per 256 positions: the three access patterns of the rounds (8 neighbouring groups x 4 columns per half wave) are conflict-free
__device__ __forceinline__ u32 mm_row(u32 pos) { return pos + (pos >> 4) + (pos >> 8); }

struct RecK { u32 k16, k24, eps; };
__device__ __forceinline__ RecK rec_consts()
{
    RecK k{65536u, 0x1000000u, 0xFFFFFFFFu};
    asm("" : "+s"(k.k16), "+s"(k.k24), "+s"(k.eps));      // (kept opaque: the compiler would turn the products into shift + 64-bit add chains)
    return k;
}
// digits d[0..7] (each < 2^23) at weights 2^(8m) -> any 64-bit representative of their sum mod p; (a0, a1): + a0 + a1 2^32
__device__ __forceinline__ u64 recombine(const int* d, const RecK& K, u32 a0, u32 a1)
{
    u64 W0 = ((u32)d[1] << 8) + (u32)d[0];
    W0 = (u64)(u32)d[2] * K.k16 + W0;
    asm("" : "+v"(W0));                                                             // (or the compiler adds the low digits last: one 64-bit add more)
    W0 = (u64)(u32)d[3] * K.k24 + W0;                                               // < 2^47
    u64 W1 = ((u32)d[5] << 8) + (u32)d[4];
    W1 = (u64)(u32)d[6] * K.k16 + W1;
    asm("" : "+v"(W1));
    W1 = (u64)(u32)d[7] * K.k24 + W1;
    W0 += a0; W1 += a1;
    const u64 Z = (u64)(u32)(W1 >> 32) * K.eps + W0;                                // W0 + (W1 >> 32) 2^64 (mod p), < 2^49
    u32 cy;
    const u32 hi = __builtin_addc((u32)(Z >> 32), (u32)W1, 0u, &cy);                // + (W1 mod 2^32) 2^32; a carry is worth 2^64 = EPS
    asm("" : "+v"(cy));
    return (u64)cy * K.eps + gl::pack((u32)Z, hi);
}

#ifndef XLP_MM_ABL
#define XLP_MM_ABL 0      // timing experiments only (wrong results): 1 = the matrices are loaded once, 2 = and no global data traffic, 3 = and no twiddle loads
#endif
__device__ __forceinline__ void load_mat(v4i (&A)[4][4], const v4i* __restrict__ mat, u32 idx, u32 lane)
{
    if (XLP_MM_ABL >= 1 && idx != 0) return;
    const v4i* m = mat + (size_t)idx * 1024 + lane;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) A[t][q] = m[(t * 4 + q) * 64];
}

// 32 DFT16 of a wave: lane (n = lane % 32, h = lane / 32) holds inputs j = 4q + 2h + ee at e[2q + ee] of DFT n and receives outputs
// k = 4t + 2h + u at y[2t + u].  K0: output 0 (t = 0, u = 0, lanes h = 0) gets + (k0lo, k0hi) (the caller passes zeros in the other lanes).
// next >= 0: once the last chain has been issued, A is reloaded with matrix `next`.
template <bool K0>
__device__ __forceinline__ void dft_op(v4i (&A)[4][4], const u64 (&e)[8], const v16i& C, const RecK& K, u32 k0lo, u32 k0hi, u64 (&y)[8],
                                       const v4i* __restrict__ mat = nullptr, int next = -1, u32 lane = 0)
{
    v4i b[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        b[q].x = (int)((u32)e[2 * q] ^ 0x80808080u); b[q].y = (int)((u32)(e[2 * q] >> 32) ^ 0x80808080u);
        b[q].z = (int)((u32)e[2 * q + 1] ^ 0x80808080u); b[q].w = (int)((u32)(e[2 * q + 1] >> 32) ^ 0x80808080u);
    }
    // two accumulators in turn: the digits of one chain are recombined (vector pipe) while the next chain runs (matrix pipe)
    v16i acc[2];
    auto chain = [&](int t) {
        acc[t & 1] = C;
#pragma unroll
        for (int q = 0; q < 4; q++) acc[t & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t][q], b[q], acc[t & 1], 0, 0, 0);
    };
    auto rec = [&](int t) {
        int d[16];
#pragma unroll
        for (int i = 0; i < 16; i++) d[i] = acc[t & 1][i];
        y[2 * t] = recombine(d, K, (K0 && t == 0) ? k0lo : 0u, (K0 && t == 0) ? k0hi : 0u);
        y[2 * t + 1] = recombine(d + 8, K, 0u, 0u);
    };
    chain(0);
    chain(1);
    __builtin_amdgcn_sched_barrier(0);
    rec(0);
    __builtin_amdgcn_sched_barrier(0);
    chain(2);
    __builtin_amdgcn_sched_barrier(0);
    rec(1);
    __builtin_amdgcn_sched_barrier(0);
    chain(3);
    __builtin_amdgcn_sched_barrier(0);
    if (next >= 0) load_mat(A, mat, (u32)next, lane);        // the next phase's matrix: its latency runs under what is left of this phase
    __builtin_amdgcn_sched_barrier(0);
    rec(2);
    rec(3);
}
// 32 DFT8 of a wave (w8 = 2^24; the matrix image is a 16 x 16 one whose rows / columns >= 8 are zero: A[t < 2][q < 2]): lane (n, h) holds
// inputs j = 4q + 2h + ee at e[2q + ee] and receives outputs k = 4t + 2h + u at y[2t + u]
template <bool K0>
__device__ __forceinline__ void dft8_op(v4i (&A)[4][4], const u64 (&e)[4], const v16i& C, const RecK& K, u32 k0lo, u32 k0hi, u64 (&y)[4],
                                        const v4i* __restrict__ mat = nullptr, int next = -1, u32 lane = 0)
{
    v4i b[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        b[q].x = (int)((u32)e[2 * q] ^ 0x80808080u); b[q].y = (int)((u32)(e[2 * q] >> 32) ^ 0x80808080u);
        b[q].z = (int)((u32)e[2 * q + 1] ^ 0x80808080u); b[q].w = (int)((u32)(e[2 * q + 1] >> 32) ^ 0x80808080u);
    }
    v16i acc[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        acc[t] = C;
#pragma unroll
        for (int q = 0; q < 2; q++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t][q], b[q], acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (next >= 0) load_mat(A, mat, (u32)next, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; t++) {
        int d[16];
#pragma unroll
        for (int i = 0; i < 16; i++) d[i] = acc[t][i];
        y[2 * t] = recombine(d, K, (K0 && t == 0) ? k0lo : 0u, (K0 && t == 0) ? k0hi : 0u);
        y[2 * t + 1] = recombine(d + 8, K, 0u, 0u);
    }
}

// ----------------------------------------------------------------- host: int8 images of field matrices, accumulator start digits
namespace {
typedef unsigned __int128 u128;
typedef __int128 i128;

// balanced signed byte digits (weights 2^(8m), m < 8) of the canonical constant c, or of c - p
bool balanced(i128 v, int8_t d[8])
{
    for (int m = 0; m < 8; m++) {
        int dm = (int)(((v % 256) + 256) % 256);
        if (dm >= 128) dm -= 256;
        d[m] = (int8_t)dm;
        v = (v - dm) / 256;
    }
    return v == 0;
}
void signed_digits(u64 c, int8_t d[8])
{
    if (c > hgl::P / 2) { if (balanced((i128)c - (i128)hgl::P, d)) return; if (balanced((i128)c, d)) return; }
    else { if (balanced((i128)c, d)) return; if (balanced((i128)c - (i128)hgl::P, d)) return; }
    abort();
}
// the int8 image of a 16 x 16 field matrix in the lane layout dft_op() reads (same layout as tools/ubench_dft16_mfma.hip)
void emit_matrix(const u64 M[16][16], uint8_t* out)
{
    static thread_local int8_t dig[16][16][8][8];      // [k][j][l][m]
    for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) {
        u64 c = M[k][j];
        for (int l = 0; l < 8; l++) { signed_digits(c, dig[k][j][l]); c = hgl::mul(c, 256); }
    }
    for (int tau = 0; tau < 4; tau++) for (int q = 0; q < 4; q++) for (int lane = 0; lane < 64; lane++) {
        const int rho = lane & 31, h = lane >> 5;
        const int u = rho >> 4, mhi = (rho >> 3) & 1, hh = (rho >> 2) & 1, mlo = rho & 3;
        const int k = 4 * tau + 2 * hh + u, m = 4 * mhi + mlo;
        uint8_t* o = out + (((size_t)(tau * 4 + q) * 64 + lane) * 16);
        for (int e = 0; e < 2; e++) for (int l = 0; l < 8; l++) o[8 * e + l] = (uint8_t)dig[k][4 * q + 2 * h + e][l][m];
    }
}
// digits >= 2^21 whose weighted sum is off (mod p)
void bias_digits(u64 off, int out[8])
{
    const u128 B = (u128)1 << 21;
    u128 base = 0;
    for (int m = 0; m < 8; m++) base += B << (8 * m);
    u128 tt = (base - off + hgl::P - 1) / hgl::P;
    u128 R = tt * hgl::P + off - base;
    for (int m = 0; m < 7; m++) { out[m] = (int)(B + (R & 255)); R >>= 8; }
    out[7] = (int)(B + R);
}
}  // namespace


}  // namespace xlp
