// K2 of the extendPol fast path on the MATRIX pipe (round 5): the 12-bit middle pass of ntt_lde_mid_kernel.inc with its 16-point DFTs done by
// v_mfma_i32_32x32x32_i8 and most of its twiddle multiplications folded into the matrices.
//
// Why it works.  w16 = 2^12 (ntt_common.hpp), so a DFT16 over Goldilocks is a 16 x 16 matrix of field constants applied to 16 words.  A word is its
// own eight byte limbs; limb l of input j times the constant M[k][j] contributes M[k][j] 2^(8l) mod p, a 64-bit constant whose eight SIGNED byte
// digits (weights 2^(8m)) make one column of a 128 x 128 int8 matrix (rows (k, m), columns (j, l)).  One MFMA chain (K = 128 = 4 instructions of
// K = 32) gives, for 32 DFT16 at once, the eight int32 digit sums of every output; ten vector instructions put them back together (recombine()).
// The matrix is ANY 16 x 16 field matrix, so whatever multiplies the inputs or outputs of a DFT16 by constants that do not depend on the DFT's
// position beyond what ONE WAVE handles is free:
//     r1   inverse DFT over digit 2 of the tile position, then x w4096^-(k low8)         shared matrix; the twiddle stays a vector multiplication
//     r2   inverse DFT over digit 1, then x w256^-(k low4)                               16 matrices, wave w owns low4 = w and w + 8
//     M_r  inverse DFT over digit 0, x G^(256 i) (coset scale), forward DFT over digit 0  ONE matrix per coset (two rounds of the VALU kernel)
//     r2'  x G^(16 j) w256^(j low4), forward DFT over digit 1                              16 matrices per coset, wave-owned
//     r3'  x G^j w4096^(j low8) (vector multiplication, table t3), forward DFT over digit 2  shared matrix
// 3 vector multiplications per input element at blow-up 2 where ntt_lde_mid_kernel.inc has 8, 8 rounds instead of 9, and no add / sub / shift
// butterflies at all.  Checked in tools/ubench_dft16_mfma.hip (profiles/r5/ubench_dft16_mfma.txt) before it was built.
//
// Signs and offsets.  MFMA int8 operands are signed: a word's bytes are xor-ed with 0x80, i.e. the matrix sees x - B0, B0 = 0x8080808080808080.
// (a) Rounds whose result goes straight into the next matrix round start their accumulators at digits that sum to B0 (mod p): the stored word is
//     y + B0 and the next round's xor takes it off again.  (b) Rounds fed by a vector multiplication see raw words: M (x - B0) = M x - B0 M 1, and
//     for the matrices used there (row-scaled DFTs with factor 1 in row 0) M 1 = 16 e_0: output 0 gets + 16 B0 (k0).  The accumulators start at
//     bias digits >= 2^21 (the largest |digit sum| of a general matrix), so every digit is a non-negative number below 2^23 and recombine() needs no signs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "ntt_common.hpp"
#include "ntt_lde_common.hpp"

namespace xlp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MM_INV = 0, MM_FWD = 1, MM_R2 = 2, MM_COSET0 = 18, MM_PER_COSET = 17;   // per coset: M_r, then r2'[16]
constexpr u32 MM_ROWS = 4096 + 256 + 16;                                             // LDS rows of the tile (mm_row)
constexpr u32 MM_THREADS = 512;

struct MidMfma {
    const v4i* mat;                    // [matrix][tile 4][k chunk 4][lane 64] x 16 bytes
    const u64* t3[LDE_MAX_COSETS];     // t3[r][low8 + 256 j] = G_r^j w4096^(j low8): what input j of r3' at low8 is multiplied by (applied when r2' stores it)
    int cz[8], cb[8];                  // accumulator start digits: sum = 0 / = B0 (mod p)
    u32 k0lo, k0hi;                    // 16 B0 mod p
};

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
    if (XLP_MM_ABL >= 1 && idx != MM_INV) return;
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

__global__ __launch_bounds__(MM_THREADS) void lde_mid_mfma(const LdeParams p, const MidMfma t, u32 BT)
{
    extern __shared__ u64 sm[];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 h = lane >> 5, b = (lane & 31) >> 2, v = lane & 3;
    const u32 ncc = (p.wc + 3) / 4;
    const u32 tile = xcd_tile();
    const u32 cc = tile % ncc, hi = tile / ncc;
    const u32 c = cc * 4;
    const u64 goff = (((u64)(cc >> 1) << p.L) << 3) + (cc & 1) * 4;                 // my 4 columns inside their 8-column group of a scratch matrix (lde_sidx)
    const u64* __restrict__ Ag = p.A + goff + ((u64)hi << 12) * 8;
    const u32 vv = c + v < p.wc ? v : p.wc - 1 - c;
    const bool valid = c + v < p.wc;
    const RecK K = rec_consts();
    const u32 k0lo = h == 0 ? t.k0lo : 0u, k0hi = h == 0 ? t.k0hi : 0u;
    v16i Cz, Cb;
#pragma unroll
    for (int i = 0; i < 16; i++) { Cz[i] = t.cz[i & 7]; Cb[i] = t.cb[i & 7]; }
    v4i A[4][4];
    u64 y[8];
    u64 sink = 0;

    // ---- r1: inverse DFT over digit 2 (positions low8 + 256 j), x w4096^-(k low8), global -> LDS.  All of the tile's words are requested up front.
    load_mat(A, t.mat, MM_INV, lane);
    {
        u64 e[4][8];
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            const u32 low8 = 8 * (wave + 8 * i) + b;
#pragma unroll
            for (u32 s = 0; s < 8; s++) {
                const u32 j = 4 * (s >> 1) + 2 * h + (s & 1);
                e[i][s] = XLP_MM_ABL >= 2 ? res_word((u64)(low8 + 256 * j) * 8 + vv + hi) : Ag[(u64)(low8 + 256 * j) * 8 + vv];
            }
        }
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            const u32 low8 = 8 * (wave + 8 * i) + b;
            u64 tw[8];
#pragma unroll
            for (u32 s = 0; s < 8; s++) tw[s] = XLP_MM_ABL >= 3 ? res_word(s * low8) : p.twi_12[(4 * (s >> 1) + 2 * h + (s & 1)) * low8];
            dft_op<true>(A, e[i], Cz, K, k0lo, k0hi, y, t.mat, i == 3 ? (int)(MM_R2 + wave) : -1, lane);
#pragma unroll
            for (u32 s = 0; s < 8; s++) {
                const u32 k = 4 * (s >> 1) + 2 * h + (s & 1);
                sm[mm_row(low8 + 256 * k) * 4 + v] = gl::mul(y[s], tw[s]);
            }
        }
    }
    __syncthreads();
    // ---- r2: inverse DFT over digit 1 (positions 256 high + 16 j + low4) x w256^-(k low4): the wave's two matrices, in place
#pragma unroll 1
    for (u32 s2 = 0; s2 < 2; s2++) {
        const u32 low4 = wave + 8 * s2;
#pragma unroll
        for (u32 i = 0; i < 2; i++) {
            const u32 base = 256 * (8 * i + b) + low4;
            u64 e[8];
#pragma unroll
            for (u32 s = 0; s < 8; s++) e[s] = sm[mm_row(base + 16 * (4 * (s >> 1) + 2 * h + (s & 1))) * 4 + v];
            dft_op<true>(A, e, Cb, K, k0lo, k0hi, y, t.mat, i == 1 ? (int)(s2 == 0 ? MM_R2 + wave + 8 : MM_COSET0) : -1, lane);
#pragma unroll
            for (u32 s = 0; s < 8; s++) sm[mm_row(base + 16 * (4 * (s >> 1) + 2 * h + (s & 1))) * 4 + v] = y[s];
        }
    }
    __syncthreads();
    // ---- the inputs of the digit-0 round stay in registers (+ B0) for every coset
    u64 X[4][8];
#pragma unroll
    for (u32 i = 0; i < 4; i++) {
        const u32 base = 16 * (8 * (wave + 8 * i) + b);
#pragma unroll
        for (u32 s = 0; s < 8; s++) X[i][s] = sm[mm_row(base + 4 * (s >> 1) + 2 * h + (s & 1)) * 4 + v];
    }
    const u32 ncos = 1u << p.e;
    for (u32 r = 0; r < ncos; r++) {
        u64* __restrict__ Br = p.B + (u64)r * lde_swords(p.wc, p.L) + goff + ((u64)hi << 12) * 8;
        const u64* __restrict__ t3 = t.t3[r];
        const u32 mr = MM_COSET0 + MM_PER_COSET * r;
        // ---- M_r: inverse DFT over digit 0, coset scale G^(256 i), forward DFT over digit 0: one matrix
        __syncthreads();                                                             // the tile's previous readers are done
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            const u32 base = 16 * (8 * (wave + 8 * i) + b);
            dft_op<false>(A, X[i], Cb, K, 0u, 0u, y, t.mat, i == 3 ? (int)(mr + 1 + wave) : -1, lane);
#pragma unroll
            for (u32 s = 0; s < 8; s++) sm[mm_row(base + 4 * (s >> 1) + 2 * h + (s & 1)) * 4 + v] = y[s];
        }
        __syncthreads();
        // ---- r2': x G^(16 j) w256^(j low4), forward DFT over digit 1, in place; the outputs leave multiplied by the next round's
        //      input factors t3 = G^j' w4096^(j' low8') (j' = high, low8' = 16 k + low4), whose loads run under the round's own work
#pragma unroll 1
        for (u32 s2 = 0; s2 < 2; s2++) {
            const u32 low4 = wave + 8 * s2;
#pragma unroll
            for (u32 i = 0; i < 2; i++) {
                const u32 base = 256 * (8 * i + b) + low4;
                u64 e[8], f[8];
#pragma unroll
                for (u32 s = 0; s < 8; s++) f[s] = XLP_MM_ABL >= 3 ? res_word(base + s) : t3[base + 16 * (4 * (s >> 1) + 2 * h + (s & 1))];
#pragma unroll
                for (u32 s = 0; s < 8; s++) e[s] = sm[mm_row(base + 16 * (4 * (s >> 1) + 2 * h + (s & 1))) * 4 + v];
                dft_op<false>(A, e, Cz, K, 0u, 0u, y, t.mat, i == 1 ? (int)(s2 == 0 ? mr + 1 + wave + 8 : MM_FWD) : -1, lane);
#pragma unroll
                for (u32 s = 0; s < 8; s++) sm[mm_row(base + 16 * (4 * (s >> 1) + 2 * h + (s & 1))) * 4 + v] = gl::mul(y[s], f[s]);
            }
        }
        __syncthreads();
        // ---- r3': forward DFT over digit 2, LDS -> B_r
#pragma unroll 1
        for (u32 i = 0; i < 4; i++) {
            const u32 low8 = 8 * (wave + 8 * i) + b;
            u64 e[8];
#pragma unroll
            for (u32 s = 0; s < 8; s++) e[s] = sm[mm_row(low8 + 256 * (4 * (s >> 1) + 2 * h + (s & 1))) * 4 + v];
            dft_op<true>(A, e, Cz, K, k0lo, k0hi, y, t.mat, (i == 3 && r + 1 < ncos) ? (int)(mr + MM_PER_COSET) : -1, lane);
            if (XLP_MM_ABL >= 2) {
#pragma unroll
                for (u32 s = 0; s < 8; s++) sink ^= gl::canon(y[s]) + low8;
            } else if (valid) {
#pragma unroll
                for (u32 s = 0; s < 8; s++) Br[(u64)(low8 + 256 * (4 * (s >> 1) + 2 * h + (s & 1))) * 8 + v] = gl::canon(y[s]);
            }
        }
    }
    if (XLP_MM_ABL >= 2) res_sink(p.A, sink);
    (void)BT; (void)sink;
}

// ----------------------------------------------------------------- host: matrices, offsets, tables
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

static int mid_mfma_tables(Ctx* ctx, u32 BT, u32 e, const u64* gk, MidMfma* out)
{
    const u32 ncos = 1u << e;
    const size_t nmat = MM_COSET0 + (size_t)MM_PER_COSET * ncos;
    const size_t mat_bytes = nmat * 16384, t3_bytes = (size_t)ncos * 4096 * 8;
    auto key = std::make_tuple(300u + 32 * e + BT, gk[0], (u64)3);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        u64* d = nullptr;
        XLP_ALLOC(plan_malloc((void**)&d, mat_bytes + t3_bytes));
        if (!plan_active()) {
            std::vector<uint8_t> hst(mat_bytes + t3_bytes);
            const u64 w12 = host_w(12), w12i = hgl::inv(w12), w16 = hgl::pow(w12, 256), w16i = hgl::inv(w16);
            if (w16 != 4096) return xlp_fail("lde mfma: w(4) is not 2^12");
            u64 M[16][16];
            for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) M[k][j] = hgl::pow(w16i, (u64)(j * k));
            emit_matrix(M, &hst[(size_t)MM_INV * 16384]);
            for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) M[k][j] = hgl::pow(w16, (u64)(j * k));
            emit_matrix(M, &hst[(size_t)MM_FWD * 16384]);
            for (int low = 0; low < 16; low++) {
                for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++)
                    M[k][j] = hgl::mul(hgl::pow(w12i, (u64)(k * low) << 4), hgl::pow(w16i, (u64)(j * k)));
                emit_matrix(M, &hst[(size_t)(MM_R2 + low) * 16384]);
            }
            for (u32 r = 0; r < ncos; r++) {
                const u64 G = gk[r], g16 = hgl::pow(G, 16), g256 = hgl::pow(G, 256);
                for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) {
                    u64 s = 0;
                    for (int i = 0; i < 16; i++) s = hgl::add(s, hgl::mul(hgl::mul(hgl::pow(w16, (u64)(i * k)), hgl::pow(g256, (u64)i)), hgl::pow(w16i, (u64)(i * j))));
                    M[k][j] = s;
                }
                emit_matrix(M, &hst[(size_t)(MM_COSET0 + MM_PER_COSET * r) * 16384]);
                for (int low = 0; low < 16; low++) {
                    for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++)
                        M[k][j] = hgl::mul(hgl::mul(hgl::pow(w16, (u64)(j * k)), hgl::pow(g16, (u64)j)), hgl::pow(w12, (u64)(j * low) << 4));
                    emit_matrix(M, &hst[(size_t)(MM_COSET0 + MM_PER_COSET * r + 1 + low) * 16384]);
                }
                u64* t3 = (u64*)&hst[mat_bytes + (size_t)r * 4096 * 8];
                for (u32 j = 0; j < 16; j++) {
                    const u64 gj = hgl::pow(G, j), wj = hgl::pow(w12, j);
                    u64 f = gj;                                      // G^j w4096^(j low8), low8 = 0
                    for (u32 low8 = 0; low8 < 256; low8++) { t3[low8 + 256 * j] = f; f = hgl::mul(f, wj); }
                }
            }
            XLP_HIP(hipMemcpyAsync(d, hst.data(), hst.size(), hipMemcpyHostToDevice, ctx->stream));
            XLP_HIP(hipStreamSynchronize(ctx->stream));
        }
        it = ctx->tables.emplace(key, d).first;
    }
    const uint8_t* base = (const uint8_t*)it->second;
    out->mat = (const v4i*)base;
    for (u32 r = 0; r < ncos; r++) out->t3[r] = (const u64*)(base + mat_bytes) + (size_t)r * 4096;
    const u64 B0 = 0x8080808080808080ull % hgl::P;
    bias_digits(0, out->cz);
    bias_digits(B0, out->cb);
    const u64 k0 = hgl::mul(16, B0);
    out->k0lo = (u32)k0; out->k0hi = (u32)(k0 >> 32);
    return 0;
}

// gk[r] = g_r^(2^BT), the coset generators of extend_fast_cols (ntt_lde.hip)
int launch_lde_mid_mfma(Ctx* ctx, const LdeParams& p, u32 BT, const u64* gk)
{
    MidMfma t{};
    if (mid_mfma_tables(ctx, BT, p.e, gk, &t)) return -1;
    static bool attr = false;
    if (!attr) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_mid_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const u32 ncc = (p.wc + 3) / 4;
    const u64 grid = (u64)ncc << BT;
    const size_t lds = (size_t)MM_ROWS * 4 * sizeof(u64);
    hipLaunchKernelGGL(lde_mid_mfma, dim3((unsigned)grid), dim3(MM_THREADS), lds, ctx->stream, p, t, BT);
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
