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

namespace xlp {

constexpr int LDE_MAX_COSETS = 8;

struct LdeParams {
    const u64* in;  u64 ld_in;     // input matrix [n][ld_in]
    u64* out;       u64 ld_out;    // output matrix [n << e][ld_out]
    u64* A;                        // scratch [n][wc]
    u64* B;                        // scratch [2^e][n][wc]
    u32 col0, wc;                  // column chunk [col0, col0 + wc)
    u32 L, e;
    const u64 *twi_bt, *twf_bt;    // in-tile twiddles of the top pass: w(BT)^-i / w(BT)^i
    const u64 *twi_12, *twf_12;    // in-tile twiddles of the 12-bit pass
    const u64 *cri_lo, *cri_hi;    // two-level powers of w(L)^-1
    const u64 *crf_lo, *crf_hi;    // two-level powers of w(L)
    const u64* sk2[LDE_MAX_COSETS];  // sk2[r][k2] = (g_r^(2^BT))^k2, k2 < 4096
    const u64* ur[LDE_MAX_COSETS];   // ur[r][j]   = g_r^j / N,       j  < 2^BT
};

template <int B>
__device__ __forceinline__ u32 tile_natural_c(u32 tpos)
{
    constexpr int m = (B + 3) / 4, dtop = B - 4 * (m - 1);
    if constexpr (m == 1) return tpos;
    else if constexpr (m == 2) return (tpos >> 4) | ((tpos & 15) << dtop);
    else return (tpos >> 8) | (((tpos >> 4) & 15) << dtop) | ((tpos & 15) << (dtop + 4));
}

// blocks b, b+8, b+16.. share an XCD and are co-resident: hand them consecutive tiles
__device__ __forceinline__ u32 xcd_tile()
{
    const u32 bid = blockIdx.x, g = gridDim.x;
    return (g >= 16 && (g & 7) == 0) ? (bid & 7) * (g >> 3) + (bid >> 3) : bid;
}

// One radix-2^D round over the tile digit at bit offset TP (compile-time geometry).
template <int B, int V, int TP, int D, bool DIT, bool INV, class Load, class Store>
__device__ __forceinline__ void fast_round(const u64* __restrict__ tw, Load load, Store store)
{
    constexpr int RAD = 1 << D;
    constexpr u32 VB = V == 8 ? 3 : 2;
    constexpr u32 NG = (1u << (B - D)) * V;
    constexpr u32 TSHIFT = B - TP - D;
    for (u32 g = threadIdx.x; g < NG; g += blockDim.x) {
        const u32 v = g & (V - 1);
        const u32 rest = g >> VB;
        const u32 low = rest & ((1u << TP) - 1);
        const u32 high = rest >> TP;
        const u32 base = (high << (TP + D)) | low;
        u64 x[RAD];
#pragma unroll
        for (int j = 0; j < RAD; j++) {
            u64 val = load(base + ((u32)j << TP), v);
            if constexpr (DIT && TP > 0) {
                if (j) val = gl::mulc(val, tw[((u32)j * low) << TSHIFT]);
            }
            x[j] = val;
        }
        dft_regs<D>(x);
#pragma unroll
        for (int k = 0; k < RAD; k++) {
            u64 val = x[brev<D>(INV ? (RAD - k) % RAD : k)];
            if constexpr (!DIT && TP > 0) {
                if (k) val = gl::mulc(val, tw[((u32)k * low) << TSHIFT]);
            }
            store(base + ((u32)k << TP), v, val);
        }
    }
}

__device__ __forceinline__ u64 two_level12(const u64* __restrict__ lo, const u64* __restrict__ hi, u64 e)
{
    const u64 t = lo[e & 4095];
    const u64 eh = e >> 12;
    return eh ? gl::mul(t, hi[eh]) : t;
}

// ------------------------------------------------------------------------ K1
template <int BT>
__global__ __launch_bounds__(1024) void lde_top_dif(const LdeParams p)
{
    constexpr int V = 8, m = (BT + 3) / 4, dtop = BT - 4 * (m - 1);
    extern __shared__ u64 smem[];
    u64* sm = smem;                                        // tile
    u64* cross = smem + ((1u << BT) + (1u << BT) / 16 + 1) * V;  // cross[k1], natural k1
    const u32 ncc = (p.wc + V - 1) / V;
    const u32 tile = xcd_tile();
    const u32 cc = tile % ncc, lo = tile / ncc;
    for (u32 k = threadIdx.x; k < (1u << BT); k += blockDim.x) cross[k] = gl::canon(two_level12(p.cri_lo, p.cri_hi, (u64)k * lo));
    const u32 c = cc * V;
    auto gload = [&](u32 tpos, u32 v) -> u64 {
        if (c + v >= p.wc) return 0;
        const u64 row = ((u64)tpos << 12) | lo;
        return gl::canon(p.in[row * p.ld_in + p.col0 + c + v]);
    };
    auto lload = [&](u32 tpos, u32 v) -> u64 { return sm[lds_row(tpos) * V + v]; };
    auto lstore = [&](u32 tpos, u32 v, u64 val) { sm[lds_row(tpos) * V + v] = val; };
    auto gstore = [&](u32 tpos, u32 v, u64 val) {
        if (c + v >= p.wc) return;
        val = gl::mulc(val, cross[tile_natural_c<BT>(tpos)]);
        const u64 row = ((u64)tpos << 12) | lo;
        p.A[row * p.wc + c + v] = val;
    };
    if constexpr (m == 2) {
        fast_round<BT, V, 4, dtop, false, true>(p.twi_bt, gload, lstore);
        __syncthreads();
        fast_round<BT, V, 0, 4, false, true>(p.twi_bt, lload, gstore);
    } else {
        fast_round<BT, V, 8, dtop, false, true>(p.twi_bt, gload, lstore);
        __syncthreads();
        fast_round<BT, V, 4, 4, false, true>(p.twi_bt, lload, lstore);
        __syncthreads();
        fast_round<BT, V, 0, 4, false, true>(p.twi_bt, lload, gstore);
    }
}

// ------------------------------------------------------------------------ K2
__global__ __launch_bounds__(1024) void lde_mid(const LdeParams p, u32 BT)
{
    constexpr int V = 4, B = 12;
    extern __shared__ u64 sm[];
    const u32 ncc = (p.wc + V - 1) / V;
    const u32 tile = xcd_tile();
    const u32 cc = tile % ncc, hi = tile / ncc;
    const u32 c = cc * V;
    const u64 n = 1ull << p.L;
    auto gload = [&](u32 tpos, u32 v) -> u64 {
        if (c + v >= p.wc) return 0;
        const u64 row = ((u64)hi << 12) | tpos;
        return p.A[row * p.wc + c + v];
    };
    auto lload = [&](u32 tpos, u32 v) -> u64 { return sm[lds_row(tpos) * V + v]; };
    auto lstore = [&](u32 tpos, u32 v, u64 val) { sm[lds_row(tpos) * V + v] = val; };
    // ---- inverse NTT over the low 12 bits (DIF); last round keeps its outputs in VGPRs
    fast_round<B, V, 8, 4, false, true>(p.twi_12, gload, lstore);
    __syncthreads();
    fast_round<B, V, 4, 4, false, true>(p.twi_12, lload, lstore);
    __syncthreads();
    // thread's group in a TP = 0 round (1024 groups == 1024 threads): base = (tid >> 2) << 4, v = tid & 3
    const u32 v = threadIdx.x & 3, base = (threadIdx.x >> 2) << 4;
    u64 X[16];
    {
        u64 x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = sm[lds_row(base + j) * V + v];
        dft_regs<4>(x);
#pragma unroll
        for (int k = 0; k < 16; k++) X[k] = x[brev<4>((16 - k) % 16)];   // coefficient at tile position base + k
    }
    const bool valid = c + v < p.wc;
    // ---- every coset: scale, forward NTT over the same 12 bits (DIT), -> B_r
    for (u32 r = 0; r < (1u << p.e); r++) {
        const u64* __restrict__ sk = p.sk2[r];
        u64* __restrict__ Br = p.B + (u64)r * n * p.wc;
        __syncthreads();   // previous users of the LDS tile are done
        {
            u64 x[16];
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = gl::mulc(X[j], sk[tile_natural_c<B>(base + j)]);
            dft_regs<4>(x);
#pragma unroll
            for (int k = 0; k < 16; k++) sm[lds_row(base + k) * V + v] = x[brev<4>(k)];
        }
        __syncthreads();
        fast_round<B, V, 4, 4, true, false>(p.twf_12, lload, lstore);
        __syncthreads();
        auto gstore = [&](u32 tpos, u32 vv, u64 val) {
            if (c + vv >= p.wc) return;
            const u64 row = ((u64)hi << 12) | tpos;
            Br[row * p.wc + c + vv] = val;
        };
        fast_round<B, V, 8, 4, true, false>(p.twf_12, lload, gstore);
    }
    (void)valid; (void)BT;
}

// ------------------------------------------------------------------------ K3
template <int BT>
__global__ __launch_bounds__(1024) void lde_top_dit(const LdeParams p)
{
    constexpr int V = 8, m = (BT + 3) / 4, dtop = BT - 4 * (m - 1);
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
    __syncthreads();
    const u32 c = cc * V;
    const u64 n = 1ull << p.L;
    const u64* __restrict__ Br = p.B + (u64)r * n * p.wc;
    auto gload = [&](u32 tpos, u32 v) -> u64 {
        if (c + v >= p.wc) return 0;
        const u64 row = ((u64)tpos << 12) | lo;
        return gl::mulc(Br[row * p.wc + c + v], cross[tile_natural_c<BT>(tpos)]);
    };
    auto lload = [&](u32 tpos, u32 v) -> u64 { return sm[lds_row(tpos) * V + v]; };
    auto lstore = [&](u32 tpos, u32 v, u64 val) { sm[lds_row(tpos) * V + v] = val; };
    auto gstore = [&](u32 tpos, u32 v, u64 val) {
        if (c + v >= p.wc) return;
        const u64 row = (((((u64)tpos << 12) | lo)) << p.e) | r;
        p.out[row * p.ld_out + p.col0 + c + v] = val;
    };
    if constexpr (m == 2) {
        fast_round<BT, V, 0, 4, true, false>(p.twf_bt, gload, lstore);
        __syncthreads();
        fast_round<BT, V, 4, dtop, true, false>(p.twf_bt, lload, gstore);
    } else {
        fast_round<BT, V, 0, 4, true, false>(p.twf_bt, gload, lstore);
        __syncthreads();
        fast_round<BT, V, 4, 4, true, false>(p.twf_bt, lload, lstore);
        __syncthreads();
        fast_round<BT, V, 8, dtop, true, false>(p.twf_bt, lload, gstore);
    }
}

// ----------------------------------------------------------------- host side
int get_pow_tables(Ctx* ctx, u32 L, u64 root, u64 mult, const u64** lo, const u64** hi);  // ntt.hip

template <int BT>
static int launch_top(Ctx* ctx, const LdeParams& p, bool dit)
{
    constexpr u32 V = 8;
    const u32 ncc = (p.wc + V - 1) / V;
    const u32 threads = (1u << (BT - 4)) * V < 64 ? 64 : (1u << (BT - 4)) * V;
    const size_t lds = ((size_t)((1u << BT) + (1u << BT) / 16 + 1) * V + (1u << BT)) * sizeof(u64);
    static bool attr_done = false;
    if (!attr_done) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_top_dif<BT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        XLP_HIP(hipFuncSetAttribute((const void*)lde_top_dit<BT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    if (!dit) {
        const u64 grid = (u64)ncc << 12;
        hipLaunchKernelGGL(lde_top_dif<BT>, dim3((unsigned)grid), dim3(threads), lds, ctx->stream, p);
    } else {
        const u64 grid = ((u64)ncc << 12) << p.e;
        hipLaunchKernelGGL(lde_top_dit<BT>, dim3((unsigned)grid), dim3(threads), lds, ctx->stream, p);
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

bool lde_fast_supported(u32 L, u32 e, u64 ncols)
{
    return L >= 17 && L <= 23 && e <= 3 && ncols >= 4;
}

int dev_extend_fast(Ctx* ctx, u64* out, const u64* in, u32 L, u32 e, u64 ncols)
{
    const u32 BT = L - 12;
    const u64 n = 1ull << L;
    const u32 ncos = 1u << e;
    LdeParams p{};
    p.in = in; p.ld_in = ncols; p.out = out; p.ld_out = ncols; p.L = L; p.e = e;
    const u64 wn = host_w(L), wbt = host_w(BT), w12 = host_w(12), wext = host_w(L + e);
    const u64* dummy;
    if (get_pow_tables(ctx, BT, hgl::inv(wbt), 1, &p.twi_bt, &dummy)) return -1;
    if (get_pow_tables(ctx, BT, wbt, 1, &p.twf_bt, &dummy)) return -1;
    if (get_pow_tables(ctx, 12, hgl::inv(w12), 1, &p.twi_12, &dummy)) return -1;
    if (get_pow_tables(ctx, 12, w12, 1, &p.twf_12, &dummy)) return -1;
    if (get_pow_tables(ctx, L, hgl::inv(wn), 1, &p.cri_lo, &p.cri_hi)) return -1;
    if (get_pow_tables(ctx, L, wn, 1, &p.crf_lo, &p.crf_hi)) return -1;
    const u64 ninv = hgl::inv(n % hgl::P);
    for (u32 r = 0; r < ncos; r++) {
        const u64 g = hgl::mul(7, hgl::pow(wext, r));          // shift * w_ext^r
        u64 gk = g;
        for (u32 i = 0; i < BT; i++) gk = hgl::mul(gk, gk);    // g^(2^BT)
        if (get_pow_tables(ctx, 12, gk, 1, &p.sk2[r], &dummy)) return -1;
        if (get_pow_tables(ctx, BT, g, ninv, &p.ur[r], &dummy)) return -1;
    }
    // column chunks: scratch = (1 + 2^e) * n * wc elements
    const u32 wc_max = ncols < 64 ? (u32)ncols : 64;
    u64* scratch = nullptr;
    if (ctx_scratch(ctx, (size_t)(1 + ncos) * n * wc_max * sizeof(u64), 1, &scratch)) return -1;
    p.A = scratch;
    p.B = scratch + n * wc_max;
    static bool mid_attr = false;
    if (!mid_attr) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_mid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        mid_attr = true;
    }
    for (u64 col0 = 0; col0 < ncols; col0 += wc_max) {
        p.col0 = (u32)col0;
        p.wc = (u32)(ncols - col0 < wc_max ? ncols - col0 : wc_max);
        p.B = p.A + n * p.wc;
        if (launch_top_bt(ctx, p, BT, false)) return -1;
        {
            const u32 ncc = (p.wc + 3) / 4;
            const u64 grid = (u64)ncc << BT;
            const size_t lds = (size_t)(4096 + 256 + 1) * 4 * sizeof(u64);
            hipLaunchKernelGGL(lde_mid, dim3((unsigned)grid), dim3(1024), lds, ctx->stream, p, BT);
            XLP_HIP(hipGetLastError());
        }
        if (launch_top_bt(ctx, p, BT, true)) return -1;
    }
    return 0;
}

}  // namespace xlp
