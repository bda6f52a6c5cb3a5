// Fast path of NTT_Goldilocks::extendPol (starks.cpp:53,134,215) for
// 2^17 <= N <= 2^23: three kernels per column chunk, 72 bytes of HBM traffic
// per input element (the generic pass engine in ntt.hip needs 96+).
//
//   N = 2^L, L = BT + 12.  Position p = t * 2^12 + lo  (t: top BT bits).
//   K1 lde_top_dif  : per (lo, 8 columns): inverse NTT over t (LDS tile 2^BT x 8),
//                     times the cross twiddle w_N^-(k1*lo), in -> A.
//   K2 lde_mid      : per (k1, 4 columns): inverse NTT over lo (tile 2^12 x 4);
//                     the coefficients stay in VGPRs; for every coset r of the
//                     extension: times (g_r^(2^BT))^k2, forward NTT over the same 12
//                     bits (decimation in time), -> B_r.
//   K3 lde_top_dit  : per (r, lo, 8 columns): times g_r^j1 / N and the cross
//                     twiddle w_N^(j1*lo), forward NTT over t, -> out rows (q << e) | r.
//   g_r = shift * w(L+e)^r; digit-reversed intermediate order, never materialised
//   in natural order (DIF then DIT), so no bit-reversal pass exists.
//
// Twiddles come from small device tables (L2-resident); the per-workgroup
// cross-twiddle vector is built once in LDS.  Same radix-16 in-register
// butterflies (shift-only inner twiddles) as ntt.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "ntt_common.hpp"
#include "ntt_lde_common.hpp"

namespace xlp {

// ------------------------------------------------------------------------ K1
#ifndef XLP_TOPV
#define XLP_TOPV 8
#endif
// XLP_PF (0 = off): the workgroup that will follow this one on its CU -- block id + XLP_PF resident workgroups, i.e. tile +
// XLP_PF / 8 under xcd_tile -- finds its first-round input in this XCD's L2 / the memory-side cache.  One dword per 64-byte
// segment of that tile is fetched with the LDS-direct form of the load (global_load_lds_dword) into a 256-byte dummy area
// in front of the tile (LDS offset 0, so M0 = 0): no VGPR receives data behind the compiler's back (a plain asm load into a register did, and corrupted
// registers the allocator had reused; a volatile load is issued system-scope and waited for on the spot), and the
// compiler inserts no wait for it.  Issued before the last round, whose arithmetic covers the latency.
#ifndef XLP_PF
#define XLP_PF 256
#endif
template <int BT>
__device__ __forceinline__ void prefetch_rows(const u64* base, u64 ld, u32 lo, u32 lds_off)
{
    for (u32 tpos = threadIdx.x; tpos < (1u << BT); tpos += blockDim.x)
        prefetch_touch(base + (((u64)tpos << 12) | lo) * ld, lds_off);
}

template <int BT, bool INV = true, bool RES = false>
__global__ __launch_bounds__(XLP_WGT) void lde_top_dif(const LdeParams p)
{
    u64 sink = 0; (void)sink;
    constexpr int V = XLP_TOPV, m = (BT + 3) / 4, dtop = BT - 4 * (m - 1);
    extern __shared__ u64 smem[];
    u64* sm = smem + (XLP_PF ? LDE_PF_LDS / 8 : 0);        // tile (the first 256 bytes: the prefetch's dummy target, LDS offset 0)
    u64* cross = sm + ((1u << BT) + (1u << BT) / 16 + 1) * V;  // cross[k1], natural k1
    const u32 ncc = (p.wc + V - 1) / V;
    const u32 tile = xcd_tile();
    const u32 cc = tile % ncc, lo = tile / ncc;
    for (u32 k = threadIdx.x; k < (1u << BT); k += blockDim.x) cross[k] = gl::canon(two_level12(p.cri_lo, p.cri_hi, (u64)k * lo));
    const u32 c = cc * V;
    // lanes past the chunk's last column read its last column (always in bounds) and never store
    auto gload = [&](u32 tpos, u32 v, u32) -> u64 {
        const u32 col = c + v < p.wc ? c + v : p.wc - 1;
        const u64 row = ((u64)tpos << 12) | lo;
        if constexpr (RES) return gl::canon(res_word(row * p.ld_in + p.col0 + col));
        else return gl::canon(gld<0>(&p.in[row * p.ld_in + p.col0 + col]));
    };
    auto lload = [&](u32, u32 v, u32 lrow) -> u64 { return sm[lrow * V + v]; };
    auto lstore = [&](u32, u32 v, u64 val, u32 lrow) { sm[lrow * V + v] = val; };
    u64* __restrict__ Ag = p.A + (((u64)cc << p.L) << 3);     // this workgroup's 8-column group of the scratch (lde_sidx)
    auto gstore = [&](u32 tpos, u32 v, u64 val, u32) {
        const u64 row = lde_srow(tpos, lo, BT);
        if constexpr (RES) sink ^= val + row;
        else if (c + v < p.wc) gst<1>(&Ag[row * 8 + v], val);
    };
    auto crossf = [&](u32 tpos, u32) -> u64 { return cross[tile_natural_c<BT>(tpos)]; };
    if constexpr (m == 2) {
        fast_round<BT, V, 4, dtop, false, INV>(p.twi_bt, gload, lstore);
        XLP_SYNC();
        fast_round<BT, V, 0, 4, false, INV, false, true>(p.twi_bt, lload, gstore, NoFactor{}, crossf);
    } else {
        fast_round<BT, V, 8, dtop, false, INV>(p.twi_bt, gload, lstore);
        XLP_SYNC();
        fast_round<BT, V, 4, 4, false, INV>(p.twi_bt, lload, lstore);
        XLP_SYNC();
#if XLP_PF
        if (!RES && gridDim.x >= 16 && (gridDim.x & 7) == 0 && blockIdx.x + XLP_PF < gridDim.x) {
            const u32 nt = tile + (XLP_PF >> 3);
            prefetch_rows<BT>(p.in + p.col0 + (nt % ncc) * V, p.ld_in, nt / ncc, 0u);
        }
#endif
        fast_round<BT, V, 0, 4, false, INV, false, true>(p.twi_bt, lload, gstore, NoFactor{}, crossf);
    }
    if constexpr (RES) res_sink(p.A, sink);
}

// K2 (lde_mid) lives in ntt_lde_mid.hip: it is compiled with another instruction-scheduling strategy (see csrc/Makefile)
int launch_lde_mid(Ctx* ctx, const LdeParams& p, u32 BT, bool resident = false);

// ---------------------------------------------------------- stand-alone NTT, bottom pass
// Second pass of a natural-order NTT / INTT of size 2^(BT+12): DIF over the low 12 bits of one
// (k1, 4 columns) tile of A, outputs written to their natural rows k1 + 2^BT * k2 (x final_scale).
template <bool INV>
__global__ __launch_bounds__(XLP_WGT) void ntt_bot_dif(const LdeParams p, u32 BT, u64 final_scale)
{
    constexpr int V = 4, B = 12;
    extern __shared__ u64 sm[];
    const u32 ncc = (p.wc + V - 1) / V;
    const u32 tile = xcd_tile();
    const u32 cc = tile % ncc, hi = tile / ncc;
    const u32 c = cc * V;
    // natural index of the top tile position `hi` (run-time BT: digits dtop,4,4 from the top)
    const u32 m = (BT + 3) / 4, dtop = BT - 4 * (m - 1);
    u32 k1 = hi >> (BT - dtop);
    {
        u32 shift = dtop;
        for (u32 r = 1; r < m; r++) { k1 |= ((hi >> (4 * (m - 1 - r))) & 15) << shift; shift += 4; }
    }
    const u64* __restrict__ Ag = p.A + (((u64)(cc >> 1) << p.L) << 3) + (cc & 1) * 4;   // my 4 columns inside their 8-column group (lde_sidx)
    auto gload = [&](u32 tpos, u32 v, u32) -> u64 {
        const u32 vv = c + v < p.wc ? v : p.wc - 1 - c;
        const u64 row = lde_srow(hi, tpos, BT);
        return gld<2>(&Ag[row * 8 + vv]);
    };
    auto lload = [&](u32, u32 v, u32 lrow) -> u64 { return sm[lrow * V + v]; };
    auto lstore = [&](u32, u32 v, u64 val, u32 lrow) { sm[lrow * V + v] = val; };
    auto gstore = [&](u32 tpos, u32 v, u64 val, u32) {
        const u64 row = (u64)k1 + ((u64)tile_natural_c<B>(tpos) << BT);
        if (final_scale != 1) val = gl::mulc(val, final_scale);
        if (c + v < p.wc) gst<5>(&p.out[row * p.ld_out + p.col0 + c + v], val);
    };
    fast_round<B, V, 8, 4, false, INV>(p.twi_12, gload, lstore);
    XLP_SYNC();
    fast_round<B, V, 4, 4, false, INV>(p.twi_12, lload, lstore);
    XLP_SYNC();
    fast_round<B, V, 0, 4, false, INV>(p.twi_12, lload, gstore);
}

// ------------------------------------------------------------------------ K3
template <int BT, bool RES = false>
__global__ __launch_bounds__(XLP_WGT) void lde_top_dit(const LdeParams p)
{
    u64 sink = 0; (void)sink;
    constexpr int V = XLP_TOPV, m = (BT + 3) / 4, dtop = BT - 4 * (m - 1);
    extern __shared__ u64 smem[];
    u64* sm = smem;
    u64* cross = smem + ((1u << BT) + (1u << BT) / 16 + 1) * V;  // cross[j1] = g_r^j1 / N * w^(j1*lo)
    const u32 ncc = (p.wc + V - 1) / V;
    const u32 ncos = 1u << p.e;
    const u32 tile = xcd_tile();
    const u32 r = tile % ncos, cc = (tile / ncos) % ncc, lo = tile / (ncos * ncc);
    const u64* __restrict__ ur = p.ur[r];
    for (u32 j = threadIdx.x; j < (1u << BT); j += blockDim.x)
        cross[j] = gl::mulc(two_level12(p.crf_lo, p.crf_hi, (u64)j * lo), ur[j]);
    XLP_SYNC();
    const u32 c = cc * V;
    const u64 n = 1ull << p.L;
    const u64* __restrict__ Br = p.B + (u64)r * lde_swords(p.wc, p.L) + (((u64)cc << p.L) << 3);   // coset r, this workgroup's 8-column group (lde_sidx)
    auto gload = [&](u32 tpos, u32 v, u32) -> u64 {
        const u32 vv = c + v < p.wc ? v : p.wc - 1 - c;
        const u64 row = lde_srow(tpos, lo, BT);
        if constexpr (RES) return res_word(row * 8 + vv + ((u64)r << 40));
        else return gld<4>(&Br[row * 8 + vv]);
    };
    auto crossf = [&](u32 tpos, u32) -> u64 { return cross[tile_natural_c<BT>(tpos)]; };
    auto lload = [&](u32, u32 v, u32 lrow) -> u64 { return sm[lrow * V + v]; };
    auto lstore = [&](u32, u32 v, u64 val, u32 lrow) { sm[lrow * V + v] = val; };
    auto gstore = [&](u32 tpos, u32 v, u64 val, u32) {
        const u64 row = (((((u64)tpos << 12) | lo)) << p.e) | r;
        if constexpr (RES) sink ^= val + row;
        else if (c + v < p.wc) gst<5>(&p.out[row * p.ld_out + p.col0 + c + v], val);
    };
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
    if constexpr (RES) res_sink(p.A, sink);
}

// tw2[pos] = G^nat(pos & ~15) * w(12)^(((pos >> 4) & 15) * (pos & 15) << 4),  pos < 4096
__global__ void lde_tw2_kernel(u64* out, const u64* __restrict__ sk /* G^k2 */, const u64* __restrict__ twf12)
{
    const u32 pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= 4096) return;
    const u32 nb = tile_natural_c<12>(pos & ~15u);
    const u32 jj = (pos >> 4) & 15, low = pos & 15;
    out[pos] = gl::mulc(sk[nb], twf12[(jj * low) << 4]);
}

// ----------------------------------------------------------------- host side
int get_pow_tables(Ctx* ctx, u32 L, u64 root, u64 mult, const u64** lo, const u64** hi);  // ntt.hip

// the derived table tw2 of a coset (lde_tw2_kernel), cached under (100 + BT, root = g^(2^BT), mult = 1); shared with ntt_lde_cm.hip
int lde_tw2_table(Ctx* ctx, u32 BT, u64 gk, const u64* sk, const u64* twf12, const u64** out)
{
    auto key = std::make_tuple(100u + BT, gk, (u64)1);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        u64* d = nullptr;
        XLP_ALLOC(plan_malloc((void**)&d, 4096 * sizeof(u64)));
        hipLaunchKernelGGL(lde_tw2_kernel, dim3(16), dim3(256), 0, ctx->stream, d, sk, twf12);
        XLP_HIP(hipGetLastError());
        it = ctx->tables.emplace(key, d).first;
    }
    *out = it->second;
    return 0;
}

template <int BT, bool RES = false>
static int launch_top(Ctx* ctx, const LdeParams& p, bool dit)
{
    constexpr u32 V = XLP_TOPV;
    const u32 ncc = (p.wc + V - 1) / V;
    u32 threads = (1u << (BT - 4)) * V < 64 ? 64 : (1u << (BT - 4)) * V;
    if (threads > XLP_WGT) threads = XLP_WGT;
    const size_t lds = ((size_t)((1u << BT) + (1u << BT) / 16 + 1) * V + (1u << BT)) * sizeof(u64) + (XLP_PF ? LDE_PF_LDS : 0);
    static bool attr_done = false;
    if (!attr_done) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_top_dif<BT, true, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        XLP_HIP(hipFuncSetAttribute((const void*)lde_top_dit<BT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    if (!dit) {
        const u64 grid = (u64)ncc << 12;
        hipLaunchKernelGGL((lde_top_dif<BT, true, RES>), dim3((unsigned)grid), dim3(threads), lds, ctx->stream, p);
    } else {
        const u64 grid = ((u64)ncc << 12) << p.e;
        hipLaunchKernelGGL((lde_top_dit<BT, RES>), dim3((unsigned)grid), dim3(threads), lds, ctx->stream, p);
    }
    XLP_HIP(hipGetLastError());
    return 0;
}

static int launch_top_bt(Ctx* ctx, const LdeParams& p, u32 BT, bool dit)
{
    switch (BT) {
        case 5: return launch_top<5>(ctx, p, dit);
        case 6: return launch_top<6>(ctx, p, dit);
        case 7: return launch_top<7>(ctx, p, dit);
        case 8: return launch_top<8>(ctx, p, dit);
        case 9: return launch_top<9>(ctx, p, dit);
        case 10: return launch_top<10>(ctx, p, dit);
        case 11: return launch_top<11>(ctx, p, dit);
        default: return xlp_fail("lde fast path: unsupported size");
    }
}

template <int BT>
static int launch_top_fwd(Ctx* ctx, const LdeParams& p)
{
    constexpr u32 V = XLP_TOPV;
    const u32 ncc = (p.wc + V - 1) / V;
    u32 threads = (1u << (BT - 4)) * V < 64 ? 64 : (1u << (BT - 4)) * V;
    if (threads > XLP_WGT) threads = XLP_WGT;
    const size_t lds = ((size_t)((1u << BT) + (1u << BT) / 16 + 1) * V + (1u << BT)) * sizeof(u64) + (XLP_PF ? LDE_PF_LDS : 0);
    static bool attr_done = false;
    if (!attr_done) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_top_dif<BT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL((lde_top_dif<BT, false>), dim3((unsigned)((u64)ncc << 12)), dim3(threads), lds, ctx->stream, p);
    XLP_HIP(hipGetLastError());
    return 0;
}

bool ntt_fast_supported(u32 L, u64 ncols) { return L >= 17 && L <= 23 && ncols >= 1; }

// Stand-alone natural-order NTT / INTT, 2^17..2^23 points: top pass (lde_top_dif: NTT over the top BT
// bits + cross twiddle) into scratch, bottom pass (ntt_bot_dif) to natural rows.  dst may alias src.
int dev_ntt_fast(Ctx* ctx, u64* dst, const u64* src, u32 L, u64 ncols, int inverse)
{
    const u32 BT = L - 12;
    const u64 n = 1ull << L;
    LdeParams p{};
    p.in = src; p.ld_in = ncols; p.out = dst; p.ld_out = ncols; p.L = L; p.e = 0;
    u64 wn = host_w(L), wbt = host_w(BT), w12 = host_w(12);
    if (inverse) { wn = hgl::inv(wn); wbt = hgl::inv(wbt); w12 = hgl::inv(w12); }
    const u64* dummy;
    // the kernels read their tables through the "inverse" slots; direction is the INV template flag
    if (get_pow_tables(ctx, BT, wbt, 1, &p.twi_bt, &dummy)) return -1;
    if (get_pow_tables(ctx, 12, w12, 1, &p.twi_12, &dummy)) return -1;
    if (get_pow_tables(ctx, L, wn, 1, &p.cri_lo, &p.cri_hi)) return -1;
    const u32 wc_max = ncols < 64 ? (u32)ncols : 64;
    // dst may alias src and the bottom pass scatters rows: a whole chunk is transformed into scratch first,
    // and every chunk's top pass must have read src before any bottom pass writes dst -> two sweeps.
    u64* scratch = nullptr;
    const u64 nchunks = (ncols + wc_max - 1) / wc_max;
    if (ctx_scratch(ctx, (size_t)lde_swords(wc_max, L) * nchunks * sizeof(u64), 1, &scratch)) return -1;
    static bool attr = false;
    if (!attr) {
        XLP_HIP(hipFuncSetAttribute((const void*)ntt_bot_dif<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        XLP_HIP(hipFuncSetAttribute((const void*)ntt_bot_dif<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    for (int sweep = 0; sweep < 2; sweep++) {
        for (u64 ci = 0; ci < nchunks; ci++) {
            p.col0 = (u32)(ci * wc_max);
            p.wc = (u32)(ncols - p.col0 < wc_max ? ncols - p.col0 : wc_max);
            p.A = scratch + ci * lde_swords(wc_max, L);
            if (sweep == 0) {
                int rc;
                if (inverse) rc = launch_top_bt(ctx, p, BT, false);
                else {
                    switch (BT) {
                        case 5: rc = launch_top_fwd<5>(ctx, p); break;
                        case 6: rc = launch_top_fwd<6>(ctx, p); break;
                        case 7: rc = launch_top_fwd<7>(ctx, p); break;
                        case 8: rc = launch_top_fwd<8>(ctx, p); break;
                        case 9: rc = launch_top_fwd<9>(ctx, p); break;
                        case 10: rc = launch_top_fwd<10>(ctx, p); break;
                        default: rc = launch_top_fwd<11>(ctx, p); break;
                    }
                }
                if (rc) return -1;
            } else {
                const u32 ncc = (p.wc + 3) / 4;
                const u64 grid = (u64)ncc << BT;
                const size_t lds = (size_t)(4096 + 256 + 1) * 4 * sizeof(u64);
                const u64 fs = inverse ? hgl::inv(n % hgl::P) : 1;
                if (inverse) hipLaunchKernelGGL(ntt_bot_dif<true>, dim3((unsigned)grid), dim3(XLP_WGT), lds, ctx->stream, p, BT, fs);
                else hipLaunchKernelGGL(ntt_bot_dif<false>, dim3((unsigned)grid), dim3(XLP_WGT), lds, ctx->stream, p, BT, fs);
                XLP_HIP(hipGetLastError());
            }
        }
    }
    return 0;
}

bool lde_fast_supported(u32 L, u32 e, u64 ncols)
{
    return L >= 17 && L <= 23 && e <= 3 && ncols >= 4;
}

int dev_extend_fast(Ctx* ctx, u64* out, const u64* in, u32 L, u32 e, u64 ncols)
{
    return dev_extend_fast_cols(ctx, out, in, L, e, ncols, 0, ncols);
}

// columns [first, first + ncols) of row-major matrices with row stride ld (in: n rows, out: n << e rows)
static int extend_fast_cols(Ctx* ctx, u64* out, const u64* in, u32 L, u32 e, u64 ld, u64 first, u64 ncols, bool resident);
int launch_lde_mid_mfma(Ctx* ctx, const LdeParams& p, u32 BT, const u64* gk, bool resident);   // ntt_lde_mid_mfma.hip
#ifdef XLP_EXP_TOP_DIT_MFMA      // experiment builds only (tools/experiments/ntt_lde_top_dit_mfma.hip linked in): lde_top_dit on the matrix pipe at N = 2^23
int launch_lde_top_dit_mfma(Ctx* ctx, const LdeParams& p, u32 BT);
#endif
// The middle pass runs on the matrix pipe (ntt_lde_mid_mfma.hip); XLP_LDE_MFMA=0 selects the vector kernel (ntt_lde_mid_kernel.inc) instead
// (A/B: tools/experiments/lde_mid_mfma_ab.py; the two produce the same bits).
bool lde_mid_mfma_enabled()
{
    const char* s = getenv("XLP_LDE_MFMA");
    return !(s && s[0] == '0');
}
int dev_extend_fast_cols(Ctx* ctx, u64* out, const u64* in, u32 L, u32 e, u64 ld, u64 first, u64 ncols)
{
    return extend_fast_cols(ctx, out, in, L, e, ld, first, ncols, false);
}
// The resident forms of the three kernels on the geometry of an extendPol of 2^L x ncols (2^22 and 2^23 rows: the bench's sizes): same tables, same
// grids, same LDS, no global data traffic (ntt_lde_common.hpp).  `sink` (>= 1024 words) is only written if a checksum takes an impossible value.
int dev_lde_resident(Ctx* ctx, u64* sink, u32 L, u32 e, u64 ncols)
{
    if (L < 22 || L > 23 || e > 3 || ncols < 4 || ncols > 64) return xlp_fail("lde resident: 2^22 or 2^23 rows, blow-up <= 8, 4..64 columns");
    return extend_fast_cols(ctx, sink, sink, L, e, ncols, 0, ncols, true);
}
static int extend_fast_cols(Ctx* ctx, u64* out, const u64* in, u32 L, u32 e, u64 ld, u64 first, u64 ncols, bool resident)
{
    const u32 BT = L - 12;
    const u64 n = 1ull << L;
    const u32 ncos = 1u << e;
    LdeParams p{};
    p.in = in; p.ld_in = ld; p.out = out; p.ld_out = ld; p.L = L; p.e = e;
    const u64 wn = host_w(L), wbt = host_w(BT), w12 = host_w(12), wext = host_w(L + e);
    const u64* dummy;
    if (get_pow_tables(ctx, BT, hgl::inv(wbt), 1, &p.twi_bt, &dummy)) return -1;
    if (get_pow_tables(ctx, BT, wbt, 1, &p.twf_bt, &dummy)) return -1;
    if (get_pow_tables(ctx, 12, hgl::inv(w12), 1, &p.twi_12, &dummy)) return -1;
    if (get_pow_tables(ctx, 12, w12, 1, &p.twf_12, &dummy)) return -1;
    if (get_pow_tables(ctx, L, hgl::inv(wn), 1, &p.cri_lo, &p.cri_hi)) return -1;
    if (get_pow_tables(ctx, L, wn, 1, &p.crf_lo, &p.crf_hi)) return -1;
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
            for (int i = 0; i < 8; i++) g256 = hgl::mul(g256, g256);   // G^256
            if (get_pow_tables(ctx, 4, g256, 1, &p.hk[r], &dummy)) return -1;   // 16 entries: (G^256)^j
        }
        if (get_pow_tables(ctx, BT, g, ninv, &p.ur[r], &dummy)) return -1;
    }
    // column chunks: scratch = (1 + 2^e) * n * wc elements
    const u32 wc_max = ncols < 64 ? (u32)ncols : 64;
    u64* scratch = nullptr;
    if (resident) scratch = out;                             // (the sink: never dereferenced but by res_sink)
    else if (ctx_scratch(ctx, (size_t)(1 + ncos) * lde_swords(wc_max, L) * sizeof(u64), 1, &scratch)) return -1;
    p.A = scratch;
    p.B = scratch + lde_swords(wc_max, L);
    for (u64 col0 = 0; col0 < ncols; col0 += wc_max) {
        p.col0 = (u32)(first + col0);
        p.wc = (u32)(ncols - col0 < wc_max ? ncols - col0 : wc_max);
        p.B = p.A + lde_swords(p.wc, L);
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        if (ctx->profile) {
            for (int i = 0; i < 4; i++) { XLP_HIP(hipEventCreate(&ev[i])); ctx->prof_events.push_back(ev[i]); }
            XLP_HIP(hipEventRecord(ev[0], ctx->stream));
        }
        if (resident ? (BT == 11 ? launch_top<11, true>(ctx, p, false) : launch_top<10, true>(ctx, p, false)) : launch_top_bt(ctx, p, BT, false)) return -1;
        if (ctx->profile) XLP_HIP(hipEventRecord(ev[1], ctx->stream));
        if (lde_mid_mfma_enabled()) { if (launch_lde_mid_mfma(ctx, p, BT, gks, resident)) return -1; }
        else if (launch_lde_mid(ctx, p, BT, resident)) return -1;
        if (ctx->profile) XLP_HIP(hipEventRecord(ev[2], ctx->stream));
#ifdef XLP_EXP_TOP_DIT_MFMA
        if (!resident && BT == 11) { if (launch_lde_top_dit_mfma(ctx, p, BT)) return -1; } else
#endif
        if (resident ? (BT == 11 ? launch_top<11, true>(ctx, p, true) : launch_top<10, true>(ctx, p, true)) : launch_top_bt(ctx, p, BT, true)) return -1;
        if (ctx->profile) XLP_HIP(hipEventRecord(ev[3], ctx->stream));
    }
    return 0;
}

}  // namespace xlp
