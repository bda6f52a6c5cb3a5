// K3 of the extendPol fast path on the matrix pipe: lde_top_dit (ntt_lde.hip) for N = 2^23 (BT = 11: the tile position is 256 d2 + 16 d1 + d0 with a
// 3-bit d2) with its DFTs done by v_mfma_i32_32x32x32_i8 (ntt_mfma_common.hpp) and its in-tile twiddles folded into matrices that one wave owns -- the
// scheme of ntt_lde_mid_mfma.hip.  Same tile, same scratch, same output rows as lde_top_dit.
//     r1  x cross[j1] = g_r^j1 / N w_N^(j1 lo) (a vector multiplication: the factor belongs to the workgroup, it cannot sit in a shared table);
//         by d2 (wave w owns d2 = w):        diag_k(w2048^(d2 k)) F16 over d0     -- the d0-half of r3's twiddle w2048^(d2 (16 k1 + k0)); it is constant
//                                                                                    over r2's DFTs and rides through them
//     r2  by k0 (= d0; w, w + 8):            F16 diag_j(w256^(j k0)) over d1
//     r3  by k1 (= d1; w, w + 8):            F8 diag_j(w128^(j k1)) over d2 (w8 = 2^24), canonical rows (q << e) | r of the output matrix
// 1 vector multiplication per element where lde_top_dit has 3, no add / sub / shift butterflies.  Other sizes run lde_top_dit.
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

constexpr int MT_R1 = 0, MT_R2 = 8, MT_R3 = 24, MT_COUNT = 40;      // r1[d2] 8, r2[k0] 16, r3[k1] 16
constexpr u32 MT_THREADS = 512;

struct TopMfma {
    const v4i* mat;
    int cz[8], cb[8];
    u32 k0lo, k0hi;                    // 16 B0 (mod p)
};

template <int BT>
__global__ __launch_bounds__(MT_THREADS) void lde_top_dit_mfma(const LdeParams p, const TopMfma t)
{
    static_assert(BT == 11, "digits 3 | 4 | 4");
    constexpr u32 NP = 1u << BT, TROWS = NP + NP / 16 + NP / 256;
    extern __shared__ u64 smem[];
    u64* sm = smem;                                   // tile [mm_row][8]
    u64* cross = smem + (size_t)TROWS * 8;            // cross[j1] = g_r^j1 / N * w^(j1 lo), natural j1
    const u32 ncc = (p.wc + 7) / 8;
    const u32 ncos = 1u << p.e;
    const u32 tile = xcd_tile();
    const u32 r = tile % ncos, cc = (tile / ncos) % ncc, lo = tile / (ncos * ncc);
    const u64* __restrict__ ur = p.ur[r];
    for (u32 j = threadIdx.x; j < NP; j += blockDim.x) cross[j] = gl::mulc(two_level12(p.crf_lo, p.crf_hi, (u64)j * lo), ur[j]);
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 h = lane >> 5, nb = (lane & 31) >> 3, v = lane & 7;
    const u32 c = cc * 8;
    const u32 vv = c + v < p.wc ? v : p.wc - 1 - c;
    const bool valid = c + v < p.wc;
    const u64* __restrict__ Br = p.B + (u64)r * lde_swords(p.wc, p.L) + (((u64)cc << p.L) << 3);   // coset r, this workgroup's 8-column group (lde_sidx)
    const RecK K = rec_consts();
    const u32 k0lo = h == 0 ? t.k0lo : 0u, k0hi = h == 0 ? t.k0hi : 0u;
    v16i Cz, Cb;
#pragma unroll
    for (int i = 0; i < 16; i++) { Cz[i] = t.cz[i & 7]; Cb[i] = t.cb[i & 7]; }
    v4i A[4][4];
    u64 y[8];
    auto lds = [&](u32 pos) -> u32 { return mm_row(pos) * 8 + v; };
    // ---- r1: x cross, forward DFT16 over d0 (positions 256 d2 + 16 d1 + j, d2 = the wave's, a chain = 4 neighbouring d1 x 8 columns), global -> LDS
    load_mat(A, t.mat, MT_R1 + wave, lane);
    {
        u64 e[4][8];
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            const u32 base = 256 * wave + 16 * (4 * i + nb);
#pragma unroll
            for (u32 s = 0; s < 8; s++) e[i][s] = Br[lde_srow(base + 4 * (s >> 1) + 2 * h + (s & 1), lo, BT) * 8 + vv];
        }
        __syncthreads();                              // cross[] is complete
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            const u32 base = 256 * wave + 16 * (4 * i + nb);
            u64 x[8];
#pragma unroll
            for (u32 s = 0; s < 8; s++) x[s] = gl::mul(e[i][s], cross[tile_natural_c<BT>(base + 4 * (s >> 1) + 2 * h + (s & 1))]);
            dft_op<true>(A, x, Cb, K, k0lo, k0hi, y, t.mat, i == 3 ? (int)(MT_R2 + wave) : -1, lane);
#pragma unroll
            for (u32 s = 0; s < 8; s++) sm[lds(base + 4 * (s >> 1) + 2 * h + (s & 1))] = y[s];
        }
    }
    __syncthreads();
    // ---- r2: x w256^(j k0), forward DFT16 over d1 (positions 256 d2 + 16 j + k0; a chain = 4 neighbouring d2 x 8 columns), in place
#pragma unroll 1
    for (u32 s2 = 0; s2 < 2; s2++) {
        const u32 k0 = wave + 8 * s2;
#pragma unroll
        for (u32 i = 0; i < 2; i++) {
            const u32 base = 256 * (4 * i + nb) + k0;
            u64 e[8];
#pragma unroll
            for (u32 s = 0; s < 8; s++) e[s] = sm[lds(base + 16 * (4 * (s >> 1) + 2 * h + (s & 1)))];
            dft_op<false>(A, e, Cb, K, 0u, 0u, y, t.mat, i == 1 ? (int)(s2 == 0 ? MT_R2 + wave + 8 : MT_R3 + wave) : -1, lane);
#pragma unroll
            for (u32 s = 0; s < 8; s++) sm[lds(base + 16 * (4 * (s >> 1) + 2 * h + (s & 1)))] = y[s];
        }
    }
    __syncthreads();
    // ---- r3: x w128^(j k1), forward DFT8 over d2 (positions 256 j + 16 k1 + k0; a chain = 4 neighbouring k0 x 8 columns), LDS -> output rows
#pragma unroll 1
    for (u32 s2 = 0; s2 < 2; s2++) {
        const u32 k1 = wave + 8 * s2;
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            const u32 low8 = 16 * k1 + 4 * i + nb;
            u64 e[4], z[4];
#pragma unroll
            for (u32 s = 0; s < 4; s++) e[s] = sm[lds(low8 + 256 * (4 * (s >> 1) + 2 * h + (s & 1)))];
            dft8_op<false>(A, e, Cz, K, 0u, 0u, z, t.mat, (i == 3 && s2 == 0) ? (int)(MT_R3 + wave + 8) : -1, lane);
            if (valid) {
#pragma unroll
                for (u32 s = 0; s < 4; s++) {
                    const u32 tpos = low8 + 256 * (4 * (s >> 1) + 2 * h + (s & 1));
                    const u64 row = (((((u64)tpos << 12) | lo)) << p.e) | r;
                    p.out[row * p.ld_out + p.col0 + c + v] = gl::canon(z[s]);
                }
            }
        }
    }
}

static int top_mfma_tables(Ctx* ctx, u32 BT, TopMfma* out)
{
    const u64 wbt = host_w(BT);
    auto key = std::make_tuple(500u + BT, wbt, (u64)6);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        u64* d = nullptr;
        XLP_ALLOC(plan_malloc((void**)&d, (size_t)MT_COUNT * 16384));
        if (!plan_active()) {
            std::vector<uint8_t> hst((size_t)MT_COUNT * 16384);
            const u64 w16 = host_w(4), w8 = host_w(3), w256 = host_w(8), w128 = host_w(7);
            if (w16 != 4096) return xlp_fail("lde mfma: w(4) is not 2^12");
            u64 M[16][16];
            for (int d2 = 0; d2 < 8; d2++) {                  // r1[d2] = diag_k(w2048^(d2 k)) F16
                for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) M[k][j] = hgl::mul(hgl::pow(wbt, (u64)(d2 * k)), hgl::pow(w16, (u64)(j * k)));
                emit_matrix(M, &hst[(size_t)(MT_R1 + d2) * 16384]);
            }
            for (int k0 = 0; k0 < 16; k0++) {                 // r2[k0] = F16 diag_j(w256^(j k0))
                for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) M[k][j] = hgl::mul(hgl::pow(w16, (u64)(j * k)), hgl::pow(w256, (u64)(j * k0)));
                emit_matrix(M, &hst[(size_t)(MT_R2 + k0) * 16384]);
            }
            for (int k1 = 0; k1 < 16; k1++) {                 // r3[k1] = F8 diag_j(w128^(j k1)), rows / columns >= 8 zero
                for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++)
                    M[k][j] = (k < 8 && j < 8) ? hgl::mul(hgl::pow(w8, (u64)(j * k)), hgl::pow(w128, (u64)(j * k1))) : 0;
                emit_matrix(M, &hst[(size_t)(MT_R3 + k1) * 16384]);
            }
            XLP_HIP(hipMemcpyAsync(d, hst.data(), hst.size(), hipMemcpyHostToDevice, ctx->stream));
            XLP_HIP(hipStreamSynchronize(ctx->stream));
        }
        it = ctx->tables.emplace(key, d).first;
    }
    out->mat = (const v4i*)it->second;
    const u64 B0 = 0x8080808080808080ull % hgl::P;
    bias_digits(0, out->cz);
    bias_digits(B0, out->cb);
    const u64 k0 = hgl::mul(16, B0);
    out->k0lo = (u32)k0; out->k0hi = (u32)(k0 >> 32);
    return 0;
}

bool lde_top_mfma_supported(u32 BT) { return BT == 11; }

int launch_lde_top_dit_mfma(Ctx* ctx, const LdeParams& p, u32 BT)
{
    if (BT != 11) return xlp_fail("lde_top_dit_mfma: N = 2^23 only");
    TopMfma t{};
    if (top_mfma_tables(ctx, BT, &t)) return -1;
    static bool attr = false;
    if (!attr) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_top_dit_mfma<11>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const u32 ncc = (p.wc + 7) / 8;
    const u64 grid = ((u64)ncc << 12) << p.e;
    const size_t lds = ((size_t)(2048 + 128 + 8) * 8 + 2048) * sizeof(u64);
    hipLaunchKernelGGL(lde_top_dit_mfma<11>, dim3((unsigned)grid), dim3(MT_THREADS), lds, ctx->stream, p, t);
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
