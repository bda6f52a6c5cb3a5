// extendPol (NTT_Goldilocks::extendPol, starks.cpp:53,134,215) on COLUMN-MAJOR sections: the three-kernel scheme of ntt_lde.hip with the
// tile's eight "columns" turned by ninety degrees.
//
// A column of N = 2^L words is a row-major [2^BT][2^12] matrix M[t][lo] of its own; the top passes transform over t for EIGHT NEIGHBOURING lo
// of ONE column (the row-major kernels: one lo of eight neighbouring columns), so that a tile row is again one 64-byte line -- of a buffer
// whose lines are full (a row-major section hands a workgroup 64-byte pieces of rows of up to 6 KB: 20 % slower inside a proof than in the
// 8-column bench).  What changes with the turn:
//   * the cross twiddle w_N^(-+ k1 * lo) differs between the tile's columns: the workgroup's LDS vector holds it for lo0, and a table
//     E[v][k1] = w_N^(-+ k1 * v), v < 8 (128 KiB, L2-resident, the same for every workgroup) supplies the rest: one more multiplication per
//     element in each of the two top passes (+3 % vector work);
//   * the middle pass owns 4 whole columns: a column is the HIGH part of its group index (a wave works on one column, adjacent lanes on
//     adjacent words: 512 contiguous bytes per wave instruction; the row-major pass read 32-byte half lines whose other half arrived too late
//     for the L2);
//   * the scratch is column-major too, and the output is COSET-MAJOR: coset r of the extension is a contiguous block of every output column
//     (written in 64-byte lines: the natural order would interleave the cosets word by word).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "ntt_common.hpp"
#include "ntt_lde_common.hpp"
#include "ntt_lde_cm_common.hpp"

namespace xlp {


// e[k * 8 + v] = root^(k v) from the two-level power tables of root (the eight columns of a position share a 64-byte line: a wave's
// 64 lanes -- 8 columns x 8 positions -- touch 8 lines per load)
__global__ void lde_cm_etable_kernel(u64* e, const u64* __restrict__ lo, const u64* __restrict__ hi, u32 BT)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (8u << BT)) return;
    const u32 v = i & 7, k = i >> 3;
    e[i] = gl::canon(two_level12(lo, hi, (u64)k * v));
}

// ------------------------------------------------------------------------ K1: inverse NTT over t of 8 neighbouring lo of one column
template <int BT>
__global__ __launch_bounds__(XLP_WGT) void lde_cm_top_dif(const LdeCmParams p)
{
    constexpr int V = 8, m = (BT + 3) / 4, dtop = BT - 4 * (m - 1);
    extern __shared__ u64 smem[];
    u64* sm = smem + LDE_PF_LDS / 8;                        // (the first 256 bytes: the dummy target of the successor prefetch, LDS offset 0)
    u64* cross = sm + ((1u << BT) + (1u << BT) / 16 + 1) * V;
    const u32 tile = xcd_tile();
    const u32 lg = tile & 511, c = tile >> 9;               // lo group fastest: co-resident workgroups share 128-byte lines
    const u32 lo0 = lg * 8;
    for (u32 k = threadIdx.x; k < (1u << BT); k += blockDim.x) cross[k] = gl::canon(two_level12(p.cri_lo, p.cri_hi, (u64)k * lo0));
    const u64* __restrict__ col = p.in + (u64)(p.col0 + c) * p.in_cs + lo0;
    u64* __restrict__ Ac = p.A + ((u64)c << p.L) + lo0;
    const u64* __restrict__ ei = p.ei;
    auto gload = [&](u32 tpos, u32 v, u32) -> u64 { return gl::canon(col[((u64)tpos << 12) + v]); };
    auto lload = [&](u32, u32 v, u32 lrow) -> u64 { return sm[lrow * V + v]; };
    auto lstore = [&](u32, u32 v, u64 val, u32 lrow) { sm[lrow * V + v] = val; };
    auto gstore = [&](u32 tpos, u32 v, u64 val, u32) { Ac[((u64)tpos << 12) + v] = val; };
    auto crossf = [&](u32 tpos, u32 v) -> u64 {
        const u32 k1 = tile_natural_c<BT>(tpos);
        return v ? gl::mul(cross[k1], ei[k1 * 8 + v]) : cross[k1];
    };
    if constexpr (m == 2) {
        fast_round<BT, V, 4, dtop, false, true>(p.twi_bt, gload, lstore);
        XLP_SYNC();
        fast_round<BT, V, 0, 4, false, true, false, true>(p.twi_bt, lload, gstore, NoFactor{}, crossf);
    } else {
        fast_round<BT, V, 8, dtop, false, true>(p.twi_bt, gload, lstore);
        XLP_SYNC();
        fast_round<BT, V, 4, 4, false, true>(p.twi_bt, lload, lstore);
        XLP_SYNC();
        // successor prefetch (see ntt_lde.hip): the workgroup that follows this one on its CU is block id + 256 = tile + 32 under xcd_tile
        if (gridDim.x >= 16 && (gridDim.x & 7) == 0 && blockIdx.x + 256 < gridDim.x) {
            const u32 nt = tile + 32;
            const u64* q = p.in + (u64)(p.col0 + (nt >> 9)) * p.in_cs + (nt & 511) * 8;
            for (u32 tpos = threadIdx.x; tpos < (1u << BT); tpos += blockDim.x) prefetch_touch(q + ((u64)tpos << 12), 0u);
        }
        fast_round<BT, V, 0, 4, false, true, false, true>(p.twi_bt, lload, gstore, NoFactor{}, crossf);
    }
}

// K2 (lde_cm_mid) lives in ntt_lde_cm_mid.hip

// ------------------------------------------------------------------------ K3: forward NTT over t of 8 neighbouring lo of one column of one coset
template <int BT>
__global__ __launch_bounds__(XLP_WGT) void lde_cm_top_dit(const LdeCmParams p)
{
    constexpr int V = 8, m = (BT + 3) / 4, dtop = BT - 4 * (m - 1);
    extern __shared__ u64 sm[];
    u64* cross = sm + ((1u << BT) + (1u << BT) / 16 + 1) * V;
    const u32 tile = xcd_tile();
    const u32 lg = tile & 511, rest = tile >> 9;
    const u32 c = rest % p.wc, r = rest / p.wc;
    const u32 lo0 = lg * 8;
    const u64* __restrict__ ur = p.ur[r];
    for (u32 j = threadIdx.x; j < (1u << BT); j += blockDim.x) cross[j] = gl::mulc(two_level12(p.crf_lo, p.crf_hi, (u64)j * lo0), ur[j]);
    XLP_SYNC();
    const u64* __restrict__ Bc = p.B + ((((u64)r * p.wc) + c) << p.L) + lo0;
    u64* __restrict__ oc = p.out + (u64)(p.col0 + c) * p.out_cs + (u64)r * p.out_cos + lo0;
    const u64* __restrict__ ef = p.ef;
    auto gload = [&](u32 tpos, u32 v, u32) -> u64 { return Bc[((u64)tpos << 12) + v]; };
    auto crossf = [&](u32 tpos, u32 v) -> u64 {
        const u32 j1 = tile_natural_c<BT>(tpos);
        return v ? gl::mul(cross[j1], ef[j1 * 8 + v]) : cross[j1];
    };
    auto lload = [&](u32, u32 v, u32 lrow) -> u64 { return sm[lrow * V + v]; };
    auto lstore = [&](u32, u32 v, u64 val, u32 lrow) { sm[lrow * V + v] = val; };
    auto gstore = [&](u32 tpos, u32 v, u64 val, u32) { oc[((u64)tpos << 12) + v] = val; };
    if constexpr (m == 2) {
        fast_round<BT, V, 0, 4, true, false, true, false>(p.twf_bt, gload, lstore, crossf, NoFactor{});
        XLP_SYNC();
        fast_round<BT, V, 4, dtop, true, false>(p.twf_bt, lload, gstore);
    } else {
        fast_round<BT, V, 0, 4, true, false, true, false>(p.twf_bt, gload, lstore, crossf, NoFactor{});
        XLP_SYNC();
        fast_round<BT, V, 4, 4, true, false>(p.twf_bt, lload, lstore);
        XLP_SYNC();
        fast_round<BT, V, 8, dtop, true, false>(p.twf_bt, lload, gstore);
    }
}

// ----------------------------------------------------------------- host side
int get_pow_tables(Ctx* ctx, u32 L, u64 root, u64 mult, const u64** lo, const u64** hi);  // ntt.hip
int lde_tw2_table(Ctx* ctx, u32 BT, u64 gk, const u64* sk, const u64* twf12, const u64** out);   // ntt_lde.hip

template <int BT>
static int launch_cm_top(Ctx* ctx, const LdeCmParams& p, bool dit)
{
    constexpr u32 V = 8;
    u32 threads = (1u << (BT - 4)) * V < 64 ? 64 : (1u << (BT - 4)) * V;
    if (threads > XLP_WGT) threads = XLP_WGT;
    const size_t lds = ((size_t)((1u << BT) + (1u << BT) / 16 + 1) * V + (1u << BT)) * sizeof(u64) + LDE_PF_LDS;
    static bool attr_done = false;
    if (!attr_done) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_cm_top_dif<BT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        XLP_HIP(hipFuncSetAttribute((const void*)lde_cm_top_dit<BT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const u64 grid = ((u64)p.wc << 9) << (dit ? p.e : 0);
    if (!dit) hipLaunchKernelGGL(lde_cm_top_dif<BT>, dim3((unsigned)grid), dim3(threads), lds, ctx->stream, p);
    else hipLaunchKernelGGL(lde_cm_top_dit<BT>, dim3((unsigned)grid), dim3(threads), lds, ctx->stream, p);
    XLP_HIP(hipGetLastError());
    return 0;
}
static int launch_cm_top_bt(Ctx* ctx, const LdeCmParams& p, u32 BT, bool dit)
{
    switch (BT) {
        case 5: return launch_cm_top<5>(ctx, p, dit);
        case 6: return launch_cm_top<6>(ctx, p, dit);
        case 7: return launch_cm_top<7>(ctx, p, dit);
        case 8: return launch_cm_top<8>(ctx, p, dit);
        case 9: return launch_cm_top<9>(ctx, p, dit);
        case 10: return launch_cm_top<10>(ctx, p, dit);
        case 11: return launch_cm_top<11>(ctx, p, dit);
        default: return xlp_fail("lde (column-major) fast path: unsupported size");
    }
}

bool lde_mid_mfma_enabled();                                                                                          // ntt_lde.hip (XLP_LDE_MFMA)
int launch_lde_cm_mid_mfma(Ctx* ctx, const u64* A, u64* B, u32 wc, u32 L, u32 e, u32 BT, const u64* gk);   // ntt_lde_mid_mfma.hip

bool lde_cm_fast_supported(u32 L, u32 e) { return L >= 17 && L <= 23 && e <= 3; }

int dev_extend_cm_fast(Ctx* ctx, u64* out, u64 out_cstride, u64 out_cos, const u64* in, u64 in_cstride, u32 L, u32 e, u64 ncols)
{
    const u32 BT = L - 12;
    const u64 n = 1ull << L;
    const u32 ncos = 1u << e;
    LdeCmParams p{};
    p.in = in; p.in_cs = in_cstride; p.out = out; p.out_cs = out_cstride; p.out_cos = out_cos; p.L = L; p.e = e;
    const u64 wn = host_w(L), wbt = host_w(BT), w12 = host_w(12), wext = host_w(L + e);
    const u64* dummy;
    if (get_pow_tables(ctx, BT, hgl::inv(wbt), 1, &p.twi_bt, &dummy)) return -1;
    if (get_pow_tables(ctx, BT, wbt, 1, &p.twf_bt, &dummy)) return -1;
    if (get_pow_tables(ctx, 12, hgl::inv(w12), 1, &p.twi_12, &dummy)) return -1;
    if (get_pow_tables(ctx, 12, w12, 1, &p.twf_12, &dummy)) return -1;
    if (get_pow_tables(ctx, L, hgl::inv(wn), 1, &p.cri_lo, &p.cri_hi)) return -1;
    if (get_pow_tables(ctx, L, wn, 1, &p.crf_lo, &p.crf_hi)) return -1;
    for (int dir = 0; dir < 2; dir++) {                     // E tables, cached under (300 + BT | 400 + BT, root, 1)
        const u64 root = dir ? wn : hgl::inv(wn);
        auto key = std::make_tuple((dir ? 400u : 300u) + BT, root, (u64)8);
        auto it = ctx->tables.find(key);
        if (it == ctx->tables.end()) {
            u64* d = nullptr;
            XLP_ALLOC(plan_malloc((void**)&d, (size_t)(8u << BT) * sizeof(u64)));
            hipLaunchKernelGGL(lde_cm_etable_kernel, dim3(((8u << BT) + 255) / 256), dim3(256), 0, ctx->stream, d, dir ? p.crf_lo : p.cri_lo, dir ? p.crf_hi : p.cri_hi, BT);
            XLP_HIP(hipGetLastError());
            it = ctx->tables.emplace(key, d).first;
        }
        (dir ? p.ef : p.ei) = it->second;
    }
    const u64 ninv = hgl::inv(n % hgl::P);
    u64 gks[LDE_MAX_COSETS] = {};
    for (u32 r = 0; r < ncos; r++) {
        const u64 g = hgl::mul(7, hgl::pow(wext, r));          // shift * w_ext^r
        u64 gk = g;
        for (u32 i = 0; i < BT; i++) gk = hgl::mul(gk, gk);    // g^(2^BT)
        gks[r] = gk;
        const u64* sk;
        if (get_pow_tables(ctx, 12, gk, 1, &sk, &dummy)) return -1;
        if (lde_tw2_table(ctx, BT, gk, sk, p.twf_12, &p.tw2[r])) return -1;
        {
            u64 g256 = gk;
            for (int i = 0; i < 8; i++) g256 = hgl::mul(g256, g256);
            if (get_pow_tables(ctx, 4, g256, 1, &p.hk[r], &dummy)) return -1;
        }
        if (get_pow_tables(ctx, BT, g, ninv, &p.ur[r], &dummy)) return -1;
    }
    // column chunks of <= 32 columns: scratch = (1 + 2^e) * n * 32 words (6.4 GB at n = 2^23, blow-up 2; whole lines whatever the chunk width)
    u64 wc_max = 32;
    if (const char* ev = getenv("XLP_LDE_CM_CHUNK")) { const u64 v = strtoull(ev, nullptr, 10); if (v >= 4 && v <= 256) wc_max = v; }
    if (ncols < wc_max) wc_max = ncols;
    u64* scratch = nullptr;
    if (ctx_scratch(ctx, (size_t)(1 + ncos) * wc_max * n * sizeof(u64), 1, &scratch)) return -1;
    for (u64 col0 = 0; col0 < ncols; col0 += wc_max) {
        p.col0 = (u32)col0;
        p.wc = (u32)(ncols - col0 < wc_max ? ncols - col0 : wc_max);
        p.A = scratch;
        p.B = scratch + (u64)p.wc * n;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        if (ctx->profile) {
            for (int i = 0; i < 4; i++) { XLP_HIP(hipEventCreate(&ev[i])); ctx->prof_events.push_back(ev[i]); }
            XLP_HIP(hipEventRecord(ev[0], ctx->stream));
        }
        if (launch_cm_top_bt(ctx, p, BT, false)) return -1;
        if (ctx->profile) XLP_HIP(hipEventRecord(ev[1], ctx->stream));
        if (lde_mid_mfma_enabled()) { if (launch_lde_cm_mid_mfma(ctx, p.A, p.B, p.wc, p.L, p.e, BT, gks)) return -1; }
        else if (launch_lde_cm_mid(ctx, p, BT)) return -1;
        if (ctx->profile) XLP_HIP(hipEventRecord(ev[2], ctx->stream));
        if (launch_cm_top_bt(ctx, p, BT, true)) return -1;
        if (ctx->profile) XLP_HIP(hipEventRecord(ev[3], ctx->stream));
    }
    return 0;
}

}  // namespace xlp
