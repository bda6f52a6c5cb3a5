// K2 of the extendPol fast path on the MATRIX pipe (round 5): the 12-bit middle pass of ntt_lde_mid_kernel.inc / ntt_lde_cm_mid.hip with its
// 16-point DFTs done by v_mfma_i32_32x32x32_i8 and EVERY in-tile factor folded into the matrices: no vector multiplication, no add / sub / shift
// butterfly is left; the vector pipe recombines digit sums.
//
// Why it works.  w16 = 2^12 (ntt_common.hpp), so a DFT16 over Goldilocks is a 16 x 16 matrix of field constants applied to 16 words.  A word is its
// own eight byte limbs; limb l of input j times the constant M[k][j] contributes M[k][j] 2^(8l) mod p, a 64-bit constant whose eight SIGNED byte
// digits (weights 2^(8m)) make one column of a 128 x 128 int8 matrix (rows (k, m), columns (j, l)).  One MFMA chain (K = 128 = 4 instructions of
// K = 32) gives, for 32 DFT16 at once, the eight int32 digit sums of every output; ~13 vector instructions put them back together (recombine()).
// The matrix is ANY 16 x 16 field matrix, so whatever multiplies the inputs or outputs of a DFT16 by constants is free as long as ONE WAVE can keep
// the matrix: every round is organised so that the factors depend on one hexadecimal digit besides the DFT's own, and wave w owns the values w and
// w + 8 of that digit (positions of the tile: 256 d2 + 16 d1 + d0; G = g_r^(2^BT), the coset generator to the tile's stride):
//     r1  by a = d1:   diag_k(w256^-(k a)) F^-1 over d2      the other half of the twiddle w4096^-(k (16 a + b)), w4096^-(k b), is constant over the
//                                                            next round's DFTs and waits until M
//     r2  by b = d0:   diag_k(w256^-(k b)) F^-1 over d1      (the same 16 matrices as r1)
//     M   by d2:       diag_k(w4096^(d2 k)) F diag(G^(256 i)) F^-1 diag_j(w4096^-(d2 j)) over d0: the last inverse round, the coset scale and the
//                                                            first forward round in ONE matrix; takes the waiting half in, hands r3' its d0-half
//     r2' by k0 = d0:  F diag_j(G^(16 j) w256^(j k0)) over d1
//     r3' by k1 = d1:  F diag_j(G^j w256^(j k1)) over d2
// 2 + 3 x 2^e rounds (the vector kernel: 3 + 3 x 2^e) and 0 of its 2 + 3 x 2^e multiplications per element.  tests/test_mfma_model.py restates both
// round structures on integers and compares them; tests/test_lde_mfma.py compares the kernels bit for bit.  The idea was measured in isolation first
// (tools/ubench_dft16_mfma.hip); a first form that kept two of the factors as vector multiplications is tools/experiments/ntt_lde_mid_mfma_partly_folded.hip.txt.
//
// Signs and offsets.  MFMA int8 operands are signed: a word's bytes are xor-ed with 0x80, i.e. the matrix sees x - B0, B0 = 0x8080808080808080.
// Rounds whose result goes into another matrix round (all but the last) start their accumulators at digits that sum to B0 (mod p): the stored word
// is y + B0 and the next round's xor takes it off again.  r1 sees raw words: M (x - B0) = M x - B0 M 1, and for a row-scaled DFT with factor 1 in
// row 0, M 1 = 16 e_0: output 0 gets + 16 B0 (k0).  The accumulators start at bias digits >= 2^21 (the largest |digit sum| of a dense matrix), so
// every digit is a non-negative number below 2^23 and recombine() needs no signs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "ntt_common.hpp"
#include "ntt_lde_common.hpp"
#include "ntt_mfma_common.hpp"

namespace xlp {

// Matrix table: [2 unused][r1 / r2: 16][per coset: 1 unused, r2'[16]][per coset: M[16], r3'[16]] -- the second per-coset block starts at the run-time index
// MidMfma::second.  (The layout of the first, partly folded form of this kernel, kept on purpose: with one per-coset block and literal indices the register
// allocator spills 25 instead of 18 registers in the row-major kernel and the pass runs 6 % slower -- same text otherwise; measured twice.)
constexpr int MM_R = 2, MM_R2P = 19, MM_R2P_PER_COSET = 17, MM_M = 0, MM_R3 = 16, MM_SECOND_PER_COSET = 32;
constexpr u32 MM_ROWS = 4096 + 256 + 16;                                                     // LDS rows of the tile (mm_row)
constexpr u32 MM_THREADS = 512;

struct MidMfma {
    const v4i* mat;                    // [matrix][tile 4][k chunk 4][lane 64] x 16 bytes
    u32 second;                        // index of coset 0's M[0]
    int cz[8], cb[8];                  // accumulator start digits: sum = 0 / = B0 (mod p)
    u32 k0lo, k0hi;                    // 16 B0 mod p
};

// What the kernel needs to know about the scratch matrices, for both layouts:
//   row-major (ntt_lde.hip, lde_sidx): groups of 8 columns, every group a [2^L][8] matrix; a workgroup's 4 columns are half of a 64-byte line
//   column-major (ntt_lde_cm.hip): column c of A at A + (c << L), of coset r's B at B + ((r wc + c) << L)
struct MidIO {
    const u64* A; u64* B;
    u32 wc, L, e;
};

// Lanes -> DFTs.  A wave's MFMA chain transforms 32 DFT16: lane (n = lane % 32, h = lane / 32) holds 8 of the 16 words of DFT n.  For each of the
// wave's two values of the owning digit there are 16 values of the third digit x 4 columns = 64 DFTs = 2 chains (i = 0, 1):
//   row-major:    n = 8 neighbouring values x the 4 columns (adjacent lanes = adjacent columns: 32-byte pieces of 8 neighbouring rows)
//   column-major: n = all 16 values x 2 columns (adjacent lanes = adjacent words of a column: 128-byte pieces)
// The LDS image (row-major: [mm_row][4]; column-major: [4][MM_ROWS], MM_ROWS = 16 mod 32) is conflict-free for every access of both maps.
template <bool CM>
struct MidMap {
    static __device__ __forceinline__ u32 whigh(u32 i, u32 lane) { return CM ? lane & 15 : 8 * i + ((lane & 31) >> 2); }
    static __device__ __forceinline__ u32 wcol(u32 i, u32 lane) { return CM ? 2 * i + ((lane >> 4) & 1) : lane & 3; }
    static __device__ __forceinline__ u32 lds(u32 pos, u32 v) { return CM ? v * MM_ROWS + mm_row(pos) : mm_row(pos) * 4 + v; }
    // word s (< 8) of a lane's DFT over the digit at bit TP (0, 4, 8): position p0 + (4 (s >> 1) + (s & 1)) << TP, p0 = the DFT's base + (2 h << TP).  mm_row is
    // affine in that digit (no carry leaves it: 2 h + 4 (s >> 1) + (s & 1) <= 15), so the eight addresses are ONE register + literals -- as lds(pos, v) per word
    // they cost ~5 vector instructions each, a third of the pass's vector instructions
    template <int TP>
    static __device__ __forceinline__ u32 lds8(u32 a0, u32 s) { return a0 + (4 * (s >> 1) + (s & 1)) * (TP == 0 ? 1u : TP == 4 ? 17u : 273u) * (CM ? 1u : 4u); }
};

// RES: the RESIDENT form (ntt_lde_common.hpp: the ALU ceiling of bench.py's headline): the same kernel with its global DATA traffic removed -- a word of
// the tile is read from the 8 KiB the sink points at (cache-resident: the instruction stream keeps its loads; made from its index instead, the resident form
// came out of the register allocator with 51 spilled registers against the product's 18 and ran SLOWER than the product), an output word goes back
// into the LDS tile; the matrices are read as in the product.
// Every round: the wave's owning digit takes the values wave, wave + 8 (r1, M, r3' in this order, r2 and r2' in the other: the matrix a round ends
// with is the one the next round starts with when both are indexed alike); per value two chains (MidMap).  The next matrix is requested right
// behind the last chain that uses the current one (dft_op), so its latency runs under the recombination of that chain.
template <bool CM, bool RES = false>
__global__ __launch_bounds__(MM_THREADS) void lde_mid_mfma(const MidIO p, const MidMfma t, u32 BT)
{
    typedef MidMap<CM> Map;
    extern __shared__ u64 sm[];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 h = lane >> 5;
    const u32 ncc = (p.wc + 3) / 4;
    const u32 tile = xcd_tile();
    const u32 cc = tile % ncc, hi = tile / ncc;
    const u32 c = cc * 4;
    const u64 rm_off = (((u64)(cc >> 1) << p.L) << 3) + (cc & 1) * 4;                          // (lde_sidx; the row of a position: lde_srow)
    auto clampv = [&](u32 v) -> u32 { return c + v < p.wc ? v : p.wc - 1 - c; };
    auto a_word = [&](u32 pos, u32 v) -> const u64* {
        return CM ? p.A + ((u64)(c + clampv(v)) << p.L) + (((u64)hi << 12) | pos) : p.A + rm_off + lde_srow(hi, pos, BT) * 8 + clampv(v);
    };
    auto b_word = [&](u32 r, u32 pos, u32 v) -> u64* {
        return CM ? p.B + (((u64)r * p.wc + c + v) << p.L) + (((u64)hi << 12) | pos) : p.B + (u64)r * lde_swords(p.wc, p.L) + rm_off + lde_srow(hi, pos, BT) * 8 + v;
    };
    // words between positions pos and pos + 256 of a column in A / B_r (digit 2 of the position: the words of a lane's first / last DFT)
    const u64 kstride = CM ? 256 : (u64)256 << (BT + 3);            // (row-major: 2^(8 + BT) scratch rows of 8 words, whatever XLP_SROW_BITS <= 8)
    const RecK K = rec_consts();
    const u32 k0lo = h == 0 ? t.k0lo : 0u, k0hi = h == 0 ? t.k0hi : 0u;
    const u32 ncos = 1u << p.e;
    v16i Cz, Cb;
#pragma unroll
    for (int i = 0; i < 16; i++) { Cz[i] = t.cz[i & 7]; Cb[i] = t.cb[i & 7]; }
    v4i A[4][4];
    u64 y[8];
    u64 sink = 0;
    // chain (s2, i) of a W round: the wave's owning digit w = wave + 8 s2 (r1 and r3' take them in this order, r2 and r2' in the other: the matrix
    // a round ends with is the one the next round starts with when both are indexed alike), the other digit x = whigh(i), column wcol(i)
    // ---- r1: inverse DFT over digit 2 (positions 256 j + 16 a + b), x w256^-(k a); global -> LDS
    load_mat(A, t.mat, MM_R + wave, lane);
    {
        u64 e[4][8];
#pragma unroll
        for (u32 o = 0; o < 4; o++) {
            const u32 low8 = 16 * (wave + 8 * (o >> 1)) + Map::whigh(o & 1, lane), v = Map::wcol(o & 1, lane);
            const u64* a0 = a_word(low8 + 512 * h, v);
#pragma unroll
            for (u32 s = 0; s < 8; s++) {
                const u32 j = 4 * (s >> 1) + (s & 1);
                e[o][s] = RES ? p.A[((low8 + 256 * (j + 2 * h)) * 8 + v) & 1023] : a0[j * kstride];      // (RES: the sink's 1024 words, cache-resident)
            }
        }
#pragma unroll
        for (u32 o = 0; o < 4; o++) {
            const u32 low8 = 16 * (wave + 8 * (o >> 1)) + Map::whigh(o & 1, lane), v = Map::wcol(o & 1, lane);
            dft_op<true>(A, e[o], Cb, K, k0lo, k0hi, y, t.mat, o == 1 ? (int)(MM_R + wave + 8) : -1, lane);     // (after o = 3 the matrix stays: r2 starts with it)
            const u32 a0 = Map::lds(low8 + 512 * h, v);
#pragma unroll
            for (u32 s = 0; s < 8; s++) sm[Map::template lds8<8>(a0, s)] = y[s];
        }
    }
    __syncthreads();
    // ---- r2: inverse DFT over digit 1 (positions 256 d2 + 16 j + b) x w256^-(k b), in place; low4 = wave + 8 first
#pragma unroll 1
    for (u32 s2 = 0; s2 < 2; s2++) {
        const u32 low4 = wave + 8 * (1 - s2);
#pragma unroll
        for (u32 i = 0; i < 2; i++) {
            const u32 base = 256 * Map::whigh(i, lane) + low4, v = Map::wcol(i, lane);
            u64 e[8];
            const u32 a0 = Map::lds(base + 32 * h, v);
#pragma unroll
            for (u32 s = 0; s < 8; s++) e[s] = sm[Map::template lds8<4>(a0, s)];
            dft_op<false>(A, e, Cb, K, 0u, 0u, y, t.mat, i == 1 ? (int)(s2 == 0 ? MM_R + wave : t.second + MM_M + wave) : -1, lane);
#pragma unroll
            for (u32 s = 0; s < 8; s++) sm[Map::template lds8<4>(a0, s)] = y[s];
        }
    }
    __syncthreads();
    // ---- the inputs of the digit-0 round (d2 = the wave's, d1 = whigh) stay in registers (+ B0) for every coset
    u64 X[4][8];
#pragma unroll
    for (u32 o = 0; o < 4; o++) {
        const u32 base = 256 * (wave + 8 * (o >> 1)) + 16 * Map::whigh(o & 1, lane), v = Map::wcol(o & 1, lane);
        const u32 a0 = Map::lds(base + 2 * h, v);
#pragma unroll
        for (u32 s = 0; s < 8; s++) X[o][s] = sm[Map::template lds8<0>(a0, s)];
    }
    for (u32 r = 0; r < ncos; r++) {
        const u32 mf = t.second + MM_SECOND_PER_COSET * r, mr2p = MM_R2P + MM_R2P_PER_COSET * r;
        // ---- M: over digit 0 (positions 256 d2 + 16 d1 + j), the wave's d2 = wave, wave + 8
        __syncthreads();                                                             // the tile's previous readers are done
#pragma unroll
        for (u32 o = 0; o < 4; o++) {
            const u32 base = 256 * (wave + 8 * (o >> 1)) + 16 * Map::whigh(o & 1, lane), v = Map::wcol(o & 1, lane);
            dft_op<false>(A, X[o], Cb, K, 0u, 0u, y, t.mat, o == 1 ? (int)(mf + MM_M + wave + 8) : o == 3 ? (int)(mr2p + wave + 8) : -1, lane);
            const u32 a0 = Map::lds(base + 2 * h, v);
#pragma unroll
            for (u32 s = 0; s < 8; s++) sm[Map::template lds8<0>(a0, s)] = y[s];
        }
        __syncthreads();
        // ---- r2': x G^(16 j) w256^(j k0), forward DFT over digit 1, in place; k0 = wave + 8 first
#pragma unroll 1
        for (u32 s2 = 0; s2 < 2; s2++) {
            const u32 low4 = wave + 8 * (1 - s2);
#pragma unroll
            for (u32 i = 0; i < 2; i++) {
                const u32 base = 256 * Map::whigh(i, lane) + low4, v = Map::wcol(i, lane);
                u64 e[8];
                const u32 a0 = Map::lds(base + 32 * h, v);
#pragma unroll
            for (u32 s = 0; s < 8; s++) e[s] = sm[Map::template lds8<4>(a0, s)];
                dft_op<false>(A, e, Cb, K, 0u, 0u, y, t.mat, i == 1 ? (int)(s2 == 0 ? mr2p + wave : mf + MM_R3 + wave) : -1, lane);
#pragma unroll
                for (u32 s = 0; s < 8; s++) sm[Map::template lds8<4>(a0, s)] = y[s];
            }
        }
        __syncthreads();
        // ---- r3': x G^j w256^(j k1), forward DFT over digit 2 (positions 256 j + 16 k1 + k0), LDS -> B_r; k1 = wave, wave + 8
#pragma unroll 1
        for (u32 s2 = 0; s2 < 2; s2++) {
#pragma unroll
            for (u32 i = 0; i < 2; i++) {
                const u32 low8 = 16 * (wave + 8 * s2) + Map::whigh(i, lane), v = Map::wcol(i, lane);
                u64 e[8];
                const u32 a0 = Map::lds(low8 + 512 * h, v);
#pragma unroll
                for (u32 s = 0; s < 8; s++) e[s] = sm[Map::template lds8<8>(a0, s)];
                dft_op<false>(A, e, Cz, K, 0u, 0u, y, t.mat,
                              i == 1 ? (int)(s2 == 0 ? mf + MM_R3 + wave + 8 : (r + 1 < ncos ? mf + MM_SECOND_PER_COSET + MM_M + wave : -1)) : -1, lane);
                if (RES) {                                                               // (back into the tile: nobody reads it before the next barrier)
#pragma unroll
                    for (u32 s = 0; s < 8; s++) sm[Map::template lds8<8>(a0, s)] = gl::canon(y[s]);
                } else if (c + v < p.wc) {
                    u64* b0 = b_word(r, low8 + 512 * h, v);
#pragma unroll
                    for (u32 s = 0; s < 8; s++) b0[(4 * (s >> 1) + (s & 1)) * kstride] = gl::canon(y[s]);
                }
            }
        }
    }
    (void)BT; (void)sink;
}

static int mid_mfma_tables(Ctx* ctx, u32 BT, u32 e, const u64* gk, MidMfma* out)
{
    const u32 ncos = 1u << e;
    const size_t second = 18 + (size_t)MM_R2P_PER_COSET * ncos, nmat = second + (size_t)MM_SECOND_PER_COSET * ncos, mat_bytes = nmat * 16384;      // 1.8 MiB at blow-up 2
    auto key = std::make_tuple(300u + 32 * e + BT, gk[0], (u64)7);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        u64* d = nullptr;
        XLP_ALLOC(plan_malloc((void**)&d, mat_bytes));
        if (!plan_active()) {
            std::vector<uint8_t> hst(mat_bytes);
            const u64 w12 = host_w(12), w12i = hgl::inv(w12), w16 = hgl::pow(w12, 256), w16i = hgl::inv(w16);
            if (w16 != 4096) return xlp_fail("lde mfma: w(4) is not 2^12");
            u64 M[16][16];
            for (int x = 0; x < 16; x++) {                                   // r1 (x = a) and r2 (x = b): diag_k(w256^-(k x)) F^-1
                for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++)
                    M[k][j] = hgl::mul(hgl::pow(w12i, (u64)(k * x) << 4), hgl::pow(w16i, (u64)(j * k)));
                emit_matrix(M, &hst[(size_t)(MM_R + x) * 16384]);
            }
            for (u32 r = 0; r < ncos; r++) {
                const u64 G = gk[r], g16 = hgl::pow(G, 16), g256 = hgl::pow(G, 256);
                const size_t mf = second + (size_t)MM_SECOND_PER_COSET * r, mr2p = MM_R2P + (size_t)MM_R2P_PER_COSET * r;
                u64 M0[16][16];                                              // F diag(G^(256 i)) F^-1
                for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) {
                    u64 s = 0;
                    for (int i = 0; i < 16; i++) s = hgl::add(s, hgl::mul(hgl::mul(hgl::pow(w16, (u64)(i * k)), hgl::pow(g256, (u64)i)), hgl::pow(w16i, (u64)(i * j))));
                    M0[k][j] = s;
                }
                for (int d2 = 0; d2 < 16; d2++) {                             // M[d2] = diag_k(w4096^(d2 k)) M0 diag_j(w4096^-(d2 j))
                    for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++)
                        M[k][j] = hgl::mul(hgl::mul(hgl::pow(w12, (u64)(d2 * k)), M0[k][j]), hgl::pow(w12i, (u64)(d2 * j)));
                    emit_matrix(M, &hst[(mf + MM_M + d2) * 16384]);
                }
                for (int k0 = 0; k0 < 16; k0++) {                             // r2'[k0] = F diag_j(G^(16 j) w256^(j k0))
                    for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++)
                        M[k][j] = hgl::mul(hgl::mul(hgl::pow(w16, (u64)(j * k)), hgl::pow(g16, (u64)j)), hgl::pow(w12, (u64)(j * k0) << 4));
                    emit_matrix(M, &hst[(mr2p + k0) * 16384]);
                }
                for (int k1 = 0; k1 < 16; k1++) {                             // r3'[k1] = F diag_j(G^j w256^(j k1))
                    for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++)
                        M[k][j] = hgl::mul(hgl::mul(hgl::pow(w16, (u64)(j * k)), hgl::pow(G, (u64)j)), hgl::pow(w12, (u64)(j * k1) << 4));
                    emit_matrix(M, &hst[(mf + MM_R3 + k1) * 16384]);
                }
            }
            XLP_HIP(hipMemcpyAsync(d, hst.data(), hst.size(), hipMemcpyHostToDevice, ctx->stream));
            XLP_HIP(hipStreamSynchronize(ctx->stream));
        }
        it = ctx->tables.emplace(key, d).first;
    }
    out->mat = (const v4i*)it->second;
    out->second = (u32)second;
    const u64 B0 = 0x8080808080808080ull % hgl::P;
    bias_digits(0, out->cz);
    bias_digits(B0, out->cb);
    const u64 k0 = hgl::mul(16, B0);
    out->k0lo = (u32)k0; out->k0hi = (u32)(k0 >> 32);
    return 0;
}

// gk[r] = g_r^(2^BT), the coset generators of extend_fast_cols / dev_extend_cm_fast
template <bool CM, bool RES>
static int launch_mid_mfma(Ctx* ctx, const MidIO& io, u32 BT, const u64* gk)
{
    MidMfma t{};
    if (mid_mfma_tables(ctx, BT, io.e, gk, &t)) return -1;
    static bool attr = false;
    if (!attr) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_mid_mfma<CM, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const u32 ncc = (io.wc + 3) / 4;
    const u64 grid = (u64)ncc << BT;
    const size_t lds = (size_t)MM_ROWS * 4 * sizeof(u64);
    hipLaunchKernelGGL((lde_mid_mfma<CM, RES>), dim3((unsigned)grid), dim3(MM_THREADS), lds, ctx->stream, io, t, BT);
    XLP_HIP(hipGetLastError());
    return 0;
}
int launch_lde_mid_mfma(Ctx* ctx, const LdeParams& p, u32 BT, const u64* gk, bool resident)
{
    const MidIO io{p.A, p.B, p.wc, p.L, p.e};
    return resident ? launch_mid_mfma<false, true>(ctx, io, BT, gk) : launch_mid_mfma<false, false>(ctx, io, BT, gk);
}
int launch_lde_cm_mid_mfma(Ctx* ctx, const u64* A, u64* B, u32 wc, u32 L, u32 e, u32 BT, const u64* gk)
{
    const MidIO io{A, B, wc, L, e};
    return launch_mid_mfma<true, false>(ctx, io, BT, gk);
}

}  // namespace xlp
