// K3 of the extendPol fast path on the matrix pipe: lde_top_dit (ntt_lde.hip) for N = 2^23 (BT = 11: digits 4 | 4 | 3 of the top tile position) with
// its DFTs done by v_mfma_i32_32x32x32_i8 (ntt_mfma_common.hpp; the idea: ntt_lde_mid_mfma.hip).  Same tile, same scratch layout, same output rows.
//     r1  x cross[j1] = g_r^j1 / N w_N^(j1 lo) (a vector multiplication: the factor belongs to the workgroup), forward DFT16 over digit 0   shared matrix
//     r2  x w256^(j low4), forward DFT16 over digit 1                                                                                      16 matrices, wave w owns low4 = w, w + 8
//         its outputs leave multiplied by the next round's input factors w2048^(j' low8') (j' = high, low8' = 16 k + low4)
//     r3  forward DFT8 over digit 2 (w8 = 2^24), canonical rows (q << e) | r of the output matrix                                           shared matrix
// 2 vector multiplications per element where lde_top_dit has 3, no add / sub / shift butterflies.  Other sizes run lde_top_dit.
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

constexpr int MT_F16 = 0, MT_F8 = 1, MT_T2 = 2, MT_COUNT = 18;
constexpr u32 MT_THREADS = 512;
#ifndef XLP_MT_R3U
#define XLP_MT_R3U 1
#endif
#ifndef XLP_MT_ABL
#define XLP_MT_ABL 0      // timing experiments only (wrong results): bit 0 = no global loads, bit 1 = no global stores
#endif

struct TopMfma {
    const v4i* mat;
    int cz[8], cb[8];
    u32 k16lo, k16hi, k8lo, k8hi;      // 16 B0, 8 B0 (mod p)
};

template <int BT>
__global__ __launch_bounds__(MT_THREADS) void lde_top_dit_mfma(const LdeParams p, const TopMfma t)
{
    static_assert(BT == 11, "digits 4 | 4 | 3");
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
    const u32 k16lo = h == 0 ? t.k16lo : 0u, k16hi = h == 0 ? t.k16hi : 0u, k8lo = h == 0 ? t.k8lo : 0u, k8hi = h == 0 ? t.k8hi : 0u;
    v16i Cz, Cb;
#pragma unroll
    for (int i = 0; i < 16; i++) { Cz[i] = t.cz[i & 7]; Cb[i] = t.cb[i & 7]; }
    v4i A[4][4];
    u64 y[8];
    u64 sink = 0;
    auto lds = [&](u32 pos) -> u32 { return mm_row(pos) * 8 + v; };
    // ---- r1: x cross, forward DFT16 over digit 0 (positions 16 g + j), global -> LDS (+ B0: the next round is a matrix round)
    load_mat(A, t.mat, MT_F16, lane);
    {
        u64 e[4][8];
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            const u32 base = 16 * (4 * (wave + 8 * i) + nb);
#pragma unroll
            for (u32 s = 0; s < 8; s++) e[i][s] = (XLP_MT_ABL & 1) ? res_word(base + s + lo) : Br[((((u64)(base + 4 * (s >> 1) + 2 * h + (s & 1))) << 12) | lo) * 8 + vv];
        }
        __syncthreads();                              // cross[] is complete
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            const u32 base = 16 * (4 * (wave + 8 * i) + nb);
            u64 x[8];
#pragma unroll
            for (u32 s = 0; s < 8; s++) x[s] = gl::mul(e[i][s], cross[tile_natural_c<BT>(base + 4 * (s >> 1) + 2 * h + (s & 1))]);
            dft_op<true>(A, x, Cb, K, k16lo, k16hi, y, t.mat, i == 3 ? (int)(MT_T2 + wave) : -1, lane);
#pragma unroll
            for (u32 s = 0; s < 8; s++) sm[lds(base + 4 * (s >> 1) + 2 * h + (s & 1))] = y[s];
        }
    }
    __syncthreads();
    // ---- r2: x w256^(j low4), forward DFT16 over digit 1 (positions 256 high + 16 j + low4), in place; outputs x w2048^(high (16 k + low4))
#pragma unroll 1
    for (u32 s2 = 0; s2 < 2; s2++) {
        const u32 low4 = wave + 8 * s2;
#pragma unroll
        for (u32 i = 0; i < 2; i++) {
            const u32 high = 4 * i + nb, base = 256 * high + low4;
            u64 e[8], f[8];
#pragma unroll
            for (u32 s = 0; s < 8; s++) f[s] = p.twf_bt[high * (16 * (4 * (s >> 1) + 2 * h + (s & 1)) + low4)];
#pragma unroll
            for (u32 s = 0; s < 8; s++) e[s] = sm[lds(base + 16 * (4 * (s >> 1) + 2 * h + (s & 1)))];
            dft_op<false>(A, e, Cz, K, 0u, 0u, y, t.mat, i == 1 ? (int)(s2 == 0 ? MT_T2 + wave + 8 : MT_F8) : -1, lane);
#pragma unroll
            for (u32 s = 0; s < 8; s++) sm[lds(base + 16 * (4 * (s >> 1) + 2 * h + (s & 1)))] = gl::mul(y[s], f[s]);
        }
    }
    __syncthreads();
    // ---- r3: forward DFT8 over digit 2 (positions low8 + 256 j), LDS -> rows (((tpos << 12) | lo) << e) | r of the output
#pragma unroll XLP_MT_R3U
    for (u32 i = 0; i < 8; i++) {
        const u32 low8 = 4 * (wave + 8 * i) + nb;
        u64 e[4], z[4];
#pragma unroll
        for (u32 s = 0; s < 4; s++) e[s] = sm[lds(low8 + 256 * (4 * (s >> 1) + 2 * h + (s & 1)))];
        dft8_op<true>(A, e, Cz, K, k8lo, k8hi, z);
        if (XLP_MT_ABL & 2) {
#pragma unroll
            for (u32 s = 0; s < 4; s++) sink ^= gl::canon(z[s]) + low8;
        } else if (valid) {
#pragma unroll
            for (u32 s = 0; s < 4; s++) {
                const u32 tpos = low8 + 256 * (4 * (s >> 1) + 2 * h + (s & 1));
                const u64 row = (((((u64)tpos << 12) | lo)) << p.e) | r;
                p.out[row * p.ld_out + p.col0 + c + v] = gl::canon(z[s]);
            }
        }
    }
    if (XLP_MT_ABL & 2) res_sink(p.A, sink);
    (void)sink;
}

static int top_mfma_tables(Ctx* ctx, u32 BT, TopMfma* out)
{
    const u64 wbt = host_w(BT);
    auto key = std::make_tuple(500u + BT, wbt, (u64)5);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        u64* d = nullptr;
        XLP_ALLOC(plan_malloc((void**)&d, (size_t)MT_COUNT * 16384));
        if (!plan_active()) {
            std::vector<uint8_t> hst((size_t)MT_COUNT * 16384);
            const u64 w16 = host_w(4), w8 = host_w(3), w256 = host_w(8);
            if (w16 != 4096) return xlp_fail("lde mfma: w(4) is not 2^12");
            u64 M[16][16];
            for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) M[k][j] = hgl::pow(w16, (u64)(j * k));
            emit_matrix(M, &hst[(size_t)MT_F16 * 16384]);
            for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) M[k][j] = (k < 8 && j < 8) ? hgl::pow(w8, (u64)(j * k)) : 0;
            emit_matrix(M, &hst[(size_t)MT_F8 * 16384]);
            for (int low = 0; low < 16; low++) {
                for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) M[k][j] = hgl::mul(hgl::pow(w16, (u64)(j * k)), hgl::pow(w256, (u64)(j * low)));
                emit_matrix(M, &hst[(size_t)(MT_T2 + low) * 16384]);
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
    const u64 k16 = hgl::mul(16, B0), k8 = hgl::mul(8, B0);
    out->k16lo = (u32)k16; out->k16hi = (u32)(k16 >> 32); out->k8lo = (u32)k8; out->k8hi = (u32)(k8 >> 32);
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
