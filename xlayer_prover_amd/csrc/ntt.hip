// Goldilocks NTT / INTT / LDE (extendPol) for gfx950.
//
// Replaces NTT_Goldilocks::NTT / INTT / extendPol of the reference's absent
// goldilocks submodule as called from src/starkpil/starks.cpp:53,134,215
// (extendPol), :262,285,326-327 (NTT/INTT) and friProve.cpp:100-102.
//
// Design (MI355X-first, not a translation of the AVX code):
//  * data stays row-major [rows][ncols] as in the reference slab
//    (stark_info.cpp:473-482).  A workgroup owns a TILE = 2^b points of one
//    strided sub-sequence x V "vector lanes" (V adjacent columns, or adjacent
//    rows x columns when ncols < V), so every global access is a V*8-byte
//    segment; tiles that are adjacent in memory go to workgroups that run on
//    the same XCD at the same time (blockIdx -> tile swizzle), so the XCD's L2
//    sees whole lines.
//  * a size-2^L transform is split four-step style into passes of <= 12 bits
//    (2 passes up to 2^23).  Inside a pass the tile sits in LDS (<= 136 KiB)
//    and is processed by radix-16 butterflies held in VGPRs: the 64th roots of
//    unity of Goldilocks are powers of two (w(6) = 8), so a 16-point DFT needs
//    only shifts/adds; one table twiddle multiply per element per round and
//    one cross-pass twiddle per element per pass.
//  * INTT runs decimation-in-frequency (natural in, digit-reversed out) and
//    the forward NTT of the LDE runs decimation-in-time (digit-reversed in,
//    natural out), so no bit-reversal pass ever touches HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "ntt_common.hpp"

namespace xlp {
using gl::u32;
using gl::u64;

// ------------------------------------------------------------------ params
constexpr int MAX_PASSES = 4;

enum RowMode : u32 {
    ROW_POS = 0,      // row = position
    ROW_NATURAL = 1,  // row = natural transform index of the position (global digit reversal)
    ROW_COSET = 2,    // row = (position << ext_bits) | coset   (LDE output interleave)
};

struct PassParams {
    const u64* src;
    u64* dst;
    u64 ncols;         // row stride (elements) of src and dst
    u32 L;             // log2 of transform size
    u32 s, b;          // this pass transforms position bits [s, s+b)
    u32 C, R;          // vector lane v in [0,V): r = v / C (adjacent low rows), c = v % C (columns); valid if r < R
    u32 n_colchunks;   // ceil(ncols / C)
    u32 n_logroups;    // ceil(2^s / R)
    u64 ntiles;        // real tile count (grid is padded to a multiple of 8)
    u32 inverse;       // inverse 16-point DFT ordering (tables hold inverse roots)
    u32 src_mode, dst_mode;
    u32 ext_bits, coset;
    u32 np;            // pass structure of the whole transform, top -> bottom (for natural-index maps)
    u32 pass_b[MAX_PASSES];
    const u64* tw;     // in-tile twiddles: tw[i] = w(b)^i, i < 2^b            (or inverse root)
    const u64* cr_lo;  // cross-pass twiddles of root w(L): cr_lo[i] = w^i (i < 4096), cr_hi[i] = w^(i << 12)
    const u64* cr_hi;
    const u64* sc_lo;  // optional load-scale tables indexed by the natural (coefficient) index (two-level like cr_*)
    const u64* sc_hi;
    u64 final_scale;   // multiplied into every stored element if != 1 (1/n of a stand-alone INTT)
};

// natural in-tile index of tile position tpos (digits top->bottom: dtop,4,4,.. ; top digit = lowest weight)
__device__ __forceinline__ u32 tile_natural(u32 tpos, u32 b)
{
    const u32 m = (b + 3) >> 2;
    const u32 dtop = b - 4 * (m - 1);
    u32 k = tpos >> (b - dtop);  // top digit -> weight 0
    u32 shift = dtop;
    for (u32 r = 1; r < m; r++) {
        const u32 dv = (tpos >> (4 * (m - 1 - r))) & 15;
        k |= dv << shift;
        shift += 4;
    }
    return k;
}

// natural transform index held at `pos` after all DIF passes (== index expected at `pos` before DIT passes)
__device__ __forceinline__ u64 natural_index(const PassParams& p, u64 pos)
{
    u64 k = 0;
    u32 consumed = 0, shift = 0;
    for (u32 i = 0; i < p.np; i++) {
        const u32 b = p.pass_b[i];
        consumed += b;
        const u32 tpos = (u32)(pos >> (p.L - consumed)) & ((1u << b) - 1);
        k |= (u64)tile_natural(tpos, b) << shift;
        shift += b;
    }
    return k;
}

__device__ __forceinline__ u64 two_level(const u64* lo, const u64* hi, u64 e, bool has_hi)
{
    u64 t = lo[e & 4095];
    if (has_hi) {
        const u64 eh = e >> 12;
        if (eh) t = gl::mul(t, hi[eh]);
    }
    return t;
}

// --------------------------------------------------------------- tile engine
template <int V>
struct TileCtx {
    const PassParams& p;
    u64* sm;           // LDS tile: lds_row(pos) * V + v
    u64 hi, lo0;       // tile coordinates
    u32 c0;
};

// One radix-2^D round over the tile digit at bit offset tp.
//  FIRST: inputs come from global memory (+ load scale, + DIT cross twiddle)
//  LAST : outputs go to global memory (+ DIF cross twiddle, + final scale)
template <int V, int D, bool DIT, bool FIRST, bool LAST>
__device__ __forceinline__ void round_body(const TileCtx<V>& t, u32 tp)
{
    const PassParams& p = t.p;
    constexpr int RAD = 1 << D;
    constexpr u32 VB = V == 8 ? 3 : 2;
    const u32 ngroups = (1u << (p.b - D)) * V;
    const bool has_tw = tp > 0;
    const u32 tshift = p.b - tp - D;          // in-tile twiddle exponent = digit * low << tshift
    const bool has_cross = p.s > 0;
    const u32 cshift = p.L - p.s - p.b;       // cross exponent = k_tile * lo << cshift
    const bool has_hi = p.L > 12;
    for (u32 g = threadIdx.x; g < ngroups; g += blockDim.x) {
        const u32 v = g & (V - 1);
        const u32 rest = g >> VB;
        const u32 low = rest & ((1u << tp) - 1);
        const u32 high = rest >> tp;
        const u32 base = (high << (tp + D)) | low;
        const u32 r = v / p.C, c = v % p.C;
        const u64 lo = t.lo0 + r;
        const bool valid = r < p.R && lo < (1ull << p.s) && (t.c0 + c) < p.ncols;
        u64 x[RAD];
        // ---- gather (+ DIT twiddles)
#pragma unroll
        for (int j = 0; j < RAD; j++) {
            const u32 tpos = base + ((u32)j << tp);
            u64 val = 0;
            if constexpr (FIRST) {
                if (valid) {
                    const u64 pos = (t.hi << (p.s + p.b)) | ((u64)tpos << p.s) | lo;
                    const u64 row = p.src_mode == ROW_POS ? pos : natural_index(p, pos);
                    val = gl::canon(p.src[row * p.ncols + t.c0 + c]);
                    if (p.sc_lo) val = gl::mulc(val, two_level(p.sc_lo, p.sc_hi, natural_index(p, pos), has_hi));
                    if (DIT && has_cross)
                        val = gl::mulc(val, two_level(p.cr_lo, p.cr_hi, ((u64)tile_natural(tpos, p.b) * lo) << cshift, has_hi));
                }
            } else {
                val = t.sm[lds_row(tpos) * V + v];
            }
            if (DIT && has_tw && j) val = gl::mulc(val, p.tw[((u32)j * low) << tshift]);
            x[j] = val;
        }
        dft_regs<D>(x);
        // ---- scatter (+ DIF twiddles)
#pragma unroll
        for (int k = 0; k < RAD; k++) {
            // forward: X[k] = x[brev(k)];  inverse: X^-1[k] = X[(RAD-k)%RAD]
            u64 val = p.inverse ? x[brev<D>((RAD - k) % RAD)] : x[brev<D>(k)];
            if (!DIT && has_tw && k) val = gl::mulc(val, p.tw[((u32)k * low) << tshift]);
            const u32 tpos = base + ((u32)k << tp);
            if constexpr (LAST) {
                if (valid) {
                    if (!DIT && has_cross)
                        val = gl::mulc(val, two_level(p.cr_lo, p.cr_hi, ((u64)tile_natural(tpos, p.b) * lo) << cshift, has_hi));
                    if (p.final_scale != 1) val = gl::mulc(val, p.final_scale);
                    const u64 pos = (t.hi << (p.s + p.b)) | ((u64)tpos << p.s) | lo;
                    u64 row = pos;
                    if (p.dst_mode == ROW_NATURAL) row = natural_index(p, pos);
                    else if (p.dst_mode == ROW_COSET) row = (pos << p.ext_bits) | p.coset;
                    p.dst[row * p.ncols + t.c0 + c] = val;
                }
            } else {
                t.sm[lds_row(tpos) * V + v] = val;
            }
        }
    }
}

template <int V, bool DIT, bool FIRST, bool LAST>
__device__ __forceinline__ void round_dispatch(const TileCtx<V>& t, u32 d, u32 tp)
{
    switch (d) {
        case 1: round_body<V, 1, DIT, FIRST, LAST>(t, tp); break;
        case 2: round_body<V, 2, DIT, FIRST, LAST>(t, tp); break;
        case 3: round_body<V, 3, DIT, FIRST, LAST>(t, tp); break;
        default: round_body<V, 4, DIT, FIRST, LAST>(t, tp); break;
    }
}

// XCD-aware tile assignment: blocks b, b+8, b+16.. share an XCD and run
// concurrently; give them tiles that are adjacent in memory (column-chunk
// fastest, then low-row group, then hi).
template <int V>
__device__ __forceinline__ bool locate_tile(const PassParams& p, TileCtx<V>& t)
{
    const u64 bid = blockIdx.x;
    u64 tile = bid;
    if (gridDim.x >= 16) tile = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);  // gridDim.x % 8 == 0 here
    if (tile >= p.ntiles) return false;  // block-uniform: grid padding only
    const u64 cc = tile % p.n_colchunks;
    const u64 rest = tile / p.n_colchunks;
    t.c0 = (u32)(cc * p.C);
    t.lo0 = (rest % p.n_logroups) * p.R;
    t.hi = rest / p.n_logroups;
    return true;
}

template <int V, bool DIT>
__global__ __launch_bounds__(1024) void ntt_pass_kernel(const PassParams p)
{
    extern __shared__ u64 sm[];
    TileCtx<V> t{p, sm, 0, 0, 0};
    if (!locate_tile<V>(p, t)) return;
    const u32 m = (p.b + 3) / 4;
    const u32 dtop = p.b - 4 * (m - 1);
    if (m == 1) {
        round_dispatch<V, DIT, true, true>(t, dtop, 0);
        return;
    }
    if (!DIT) {
        // top digit first
        round_dispatch<V, DIT, true, false>(t, dtop, 4 * (m - 1));
        __syncthreads();
        for (u32 r = 1; r + 1 < m; r++) {
            round_body<V, 4, DIT, false, false>(t, 4 * (m - 1 - r));
            __syncthreads();
        }
        round_body<V, 4, DIT, false, true>(t, 0);
    } else {
        // bottom digit first
        round_body<V, 4, DIT, true, false>(t, 0);
        __syncthreads();
        for (u32 r = m - 2; r >= 1; r--) {
            round_body<V, 4, DIT, false, false>(t, 4 * (m - 1 - r));
            __syncthreads();
        }
        round_dispatch<V, DIT, false, true>(t, dtop, 4 * (m - 1));
    }
}

// -------------------------------------------------------------- table kernels
// out[i] = mult * base^(i << shift), i < count
__global__ void pow_table_kernel(u64* out, u64 base, u32 shift, u64 count, u64 mult)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 b = base;
    for (u32 k = 0; k < shift; k++) b = gl::mul(b, b);
    out[i] = gl::canon(gl::mul(gl::pow(b, i), mult));
}

__global__ void copy_canon_kernel(u64* dst, const u64* src, u64 count, u64 scale)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 v = gl::canon(src[i]);
    dst[i] = scale == 1 ? v : gl::mulc(v, scale);
}

__global__ void bcast_row_kernel(u64* dst, const u64* row, u64 ncols, u64 count)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = gl::canon(row[i % ncols]);
}

// ------------------------------------------------------------------ host side
static const u64 W32 = 7277203076849721926ULL;  // Goldilocks::w(32), pinned by golden proofs

u64 host_w(unsigned n)
{
    u64 w = W32;
    for (unsigned i = 32; i > n; i--) w = hgl::mul(w, w);
    return w;
}

// Two-level power tables of `root` over exponents < 2^L: [0,4096) = mult*root^i, [4096..) = root^(i<<12).
int get_pow_tables(Ctx* ctx, u32 L, u64 root, u64 mult, const u64** lo, const u64** hi)
{
    auto key = std::make_tuple(L, root, mult);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        u64 nlo = L < 12 ? (1ull << L) : 4096, nhi = L > 12 ? (1ull << (L - 12)) : 0;
        u64* d = nullptr;
        XLP_ALLOC(plan_malloc((void**)&d, (4096 + nhi) * sizeof(u64)));
        XLP_HIP(hipMemsetAsync(d, 0, (4096 + nhi) * sizeof(u64), ctx->stream));
        hipLaunchKernelGGL(pow_table_kernel, dim3((nlo + 255) / 256), dim3(256), 0, ctx->stream, d, root, 0u, nlo, mult);
        if (nhi) hipLaunchKernelGGL(pow_table_kernel, dim3((nhi + 255) / 256), dim3(256), 0, ctx->stream, d + 4096, root, 12u, nhi, (u64)1);
        XLP_HIP(hipGetLastError());
        it = ctx->tables.emplace(key, d).first;
    }
    *lo = it->second;
    *hi = L > 12 ? it->second + 4096 : nullptr;
    return 0;
}

struct Plan {
    u32 L, np;
    u32 b[MAX_PASSES], s[MAX_PASSES];
};

// passes of <= 12 bits, top -> bottom; the bottom pass is the big one (12 bits,
// V=4: its data is contiguous), upper passes <= 11 bits (V=8: 64-byte segments).
static Plan make_plan(u32 L)
{
    Plan pl{};
    pl.L = L;
    if (L <= 12) { pl.np = L ? 1 : 0; pl.b[0] = L; pl.s[0] = 0; return pl; }
    u32 rest = L - 12;
    u32 nup = (rest + 10) / 11;
    pl.np = nup + 1;
    for (u32 i = 0; i < nup; i++) pl.b[i] = rest / nup + (i < rest % nup ? 1 : 0);
    pl.b[nup] = 12;
    u32 acc = 0;
    for (u32 i = pl.np; i-- > 0;) { pl.s[i] = acc; acc += pl.b[i]; }
    return pl;
}

template <int V, bool DIT>
static int launch_pass_v(Ctx* ctx, PassParams& p)
{
    const u64 ncols = p.ncols;
    p.C = ncols < (u64)V ? (u32)ncols : V;
    u32 R = 1;
    while (R * 2 * p.C <= (u32)V && (u64)R * 2 <= (1ull << p.s)) R *= 2;
    p.R = R;
    p.n_colchunks = (u32)((ncols + p.C - 1) / p.C);
    p.n_logroups = (u32)(((1ull << p.s) + R - 1) / R);
    u64 ntiles = (u64)p.n_colchunks * p.n_logroups * (1ull << (p.L - p.s - p.b));
    u64 grid = ntiles < 16 ? ntiles : (ntiles + 7) / 8 * 8;
    if (grid > 0x7fffffffULL) return xlp_fail("ntt: grid too large");
    u32 groups = (1u << (p.b > 4 ? p.b - 4 : 0)) * V;
    u32 threads = groups < 64 ? 64 : (groups > 1024 ? 1024 : groups);
    u32 rows = 1u << p.b;
    size_t lds = (size_t)(rows + (rows >> 4) + 1) * V * sizeof(u64);
    p.ntiles = ntiles;
    hipLaunchKernelGGL((ntt_pass_kernel<V, DIT>), dim3((unsigned)grid), dim3(threads), lds, ctx->stream, p);
    XLP_HIP(hipGetLastError());
    return 0;
}

static int launch_pass(Ctx* ctx, PassParams& p, bool dit)
{
    // V=4 only for 12-bit tiles (LDS capacity); they are the bottom pass whose rows are contiguous
    if (p.b == 12) return dit ? launch_pass_v<4, true>(ctx, p) : launch_pass_v<4, false>(ctx, p);
    return dit ? launch_pass_v<8, true>(ctx, p) : launch_pass_v<8, false>(ctx, p);
}

static void fill_plan(PassParams& p, const Plan& pl)
{
    p.np = pl.np;
    for (u32 i = 0; i < pl.np; i++) p.pass_b[i] = pl.b[i];
}

int ntt_init(Ctx* ctx)
{
    XLP_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    XLP_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    XLP_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    XLP_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    (void)ctx;
    return 0;
}

// in-tile twiddle table for a pass of b bits: tw[i] = root(2^b)^i (single level, <= 4096 entries)
static int get_tile_table(Ctx* ctx, u32 b, bool inverse, const u64** tw)
{
    u64 root = host_w(b);
    if (inverse) root = hgl::inv(root);
    const u64* hi;
    return get_pow_tables(ctx, b, root, 1, tw, &hi);
}

// Stand-alone NTT / INTT, natural order in and out.  dst may alias src.
int dev_ntt(Ctx* ctx, u64* dst, const u64* src, u64 n, u64 ncols, int inverse)
{
    if (n == 0 || ncols == 0) return 0;
    if (n & (n - 1)) return xlp_fail("ntt: size is not a power of two");
    u32 L = 0;
    while ((1ull << L) < n) L++;
    if (L > 32) return xlp_fail("ntt: size exceeds 2^32");
    if (L == 0) {
        u64 cnt = ncols;
        hipLaunchKernelGGL(copy_canon_kernel, dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, dst, src, cnt, (u64)1);
        XLP_HIP(hipGetLastError());
        return 0;
    }
    if (ntt_fast_supported(L, ncols)) return dev_ntt_fast(ctx, dst, src, L, ncols, inverse);
    Plan pl = make_plan(L);
    u64 root = host_w(L);
    if (inverse) root = hgl::inv(root);
    const u64 *clo, *chi;
    if (get_pow_tables(ctx, L, root, 1, &clo, &chi)) return -1;
    u64* work = nullptr;
    if (pl.np > 1) {
        if (ctx_scratch(ctx, n * ncols * sizeof(u64), 0, &work)) return -1;
    }
    for (u32 i = 0; i < pl.np; i++) {
        PassParams p{};
        p.ncols = ncols; p.L = L; p.s = pl.s[i]; p.b = pl.b[i];
        p.inverse = inverse ? 1 : 0;
        if (get_tile_table(ctx, p.b, inverse, &p.tw)) return -1;
        p.cr_lo = clo; p.cr_hi = chi; p.sc_lo = nullptr; p.sc_hi = nullptr;
        p.final_scale = 1;
        fill_plan(p, pl);
        const bool first = i == 0, last = i == pl.np - 1;
        p.src = first ? src : work;
        p.dst = last ? dst : work;
        p.src_mode = ROW_POS;
        p.dst_mode = last ? ROW_NATURAL : ROW_POS;
        if (last && inverse) p.final_scale = hgl::inv(n % hgl::P);
        if (launch_pass(ctx, p, false)) return -1;
    }
    return 0;
}

// extendPol: out[k][c] = P_c(shift * w(log2 n_ext)^k), P_c = interpolant of in[.][c].
// scratch: >= 2*n*ncols elements of device memory (reference passes an
// out-sized buffer, starks.cpp:53); if NULL the context's own scratch is used.
int dev_extend(Ctx* ctx, u64* out, const u64* in, u64 n_ext, u64 n, u64 ncols, u64* scratch)
{
    if (n == 0 || ncols == 0) return 0;
    if ((n & (n - 1)) || (n_ext & (n_ext - 1)) || n_ext < n) return xlp_fail("extendPol: sizes must be powers of two, n_ext >= n");
    u32 L = 0, Le = 0;
    while ((1ull << L) < n) L++;
    while ((1ull << Le) < n_ext) Le++;
    if (Le > 32) return xlp_fail("extendPol: size exceeds 2^32");
    const u32 e = Le - L;
    const u64 ncoset = 1ull << e;
    const u64 ninv = hgl::inv(n % hgl::P);
    if (lde_fast_supported(L, e, ncols)) return dev_extend_fast(ctx, out, in, L, e, ncols);
    if (L == 0) {
        // constant polynomial: every output row equals the input row
        const u64 cnt = n_ext * ncols;
        hipLaunchKernelGGL(bcast_row_kernel, dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, out, in, ncols, cnt);
        XLP_HIP(hipGetLastError());
        return 0;
    }
    Plan pl = make_plan(L);
    u64 *A = nullptr, *B = nullptr;
    const u64 elems = n * ncols;
    if (scratch && e >= 1) { A = scratch; B = scratch + elems; }
    else {
        if (ctx_scratch(ctx, 2 * elems * sizeof(u64), 0, &A)) return -1;
        B = A + elems;
    }
    // ---- INTT (DIF): in -> A, coefficients left in digit-reversed positions
    const u64 wn = host_w(L);
    const u64 *ilo, *ihi, *flo, *fhi;
    if (get_pow_tables(ctx, L, hgl::inv(wn), 1, &ilo, &ihi)) return -1;
    if (get_pow_tables(ctx, L, wn, 1, &flo, &fhi)) return -1;
    for (u32 i = 0; i < pl.np; i++) {
        PassParams p{};
        p.ncols = ncols; p.L = L; p.s = pl.s[i]; p.b = pl.b[i];
        p.inverse = 1;
        if (get_tile_table(ctx, p.b, true, &p.tw)) return -1;
        p.cr_lo = ilo; p.cr_hi = ihi;
        p.final_scale = 1;
        fill_plan(p, pl);
        p.src = i == 0 ? in : A;
        p.dst = A;
        p.src_mode = ROW_POS; p.dst_mode = ROW_POS;
        if (launch_pass(ctx, p, false)) return -1;
    }
    // ---- per coset r: coefficients * (shift*w_ext^r)^i / n, then DIT NTT -> out rows (q << e) | r
    const u64 wext = host_w(Le);
    for (u64 r = 0; r < ncoset; r++) {
        const u64 g = hgl::mul(7, hgl::pow(wext, r));
        const u64 *slo, *shi;
        if (get_pow_tables(ctx, L, g, ninv, &slo, &shi)) return -1;
        for (u32 q = 0; q < pl.np; q++) {
            const u32 i = pl.np - 1 - q;  // DIT: bottom pass first
            PassParams p{};
            p.ncols = ncols; p.L = L; p.s = pl.s[i]; p.b = pl.b[i];
            p.inverse = 0;
            if (get_tile_table(ctx, p.b, false, &p.tw)) return -1;
            p.cr_lo = flo; p.cr_hi = fhi;
            p.final_scale = 1;
            fill_plan(p, pl);
            const bool first = q == 0, last = q == pl.np - 1;
            p.src = first ? A : B;
            p.dst = last ? out : B;
            p.src_mode = ROW_POS;
            p.dst_mode = last ? ROW_COSET : ROW_POS;
            p.ext_bits = e; p.coset = (u32)r;
            if (first) { p.sc_lo = slo; p.sc_hi = shi; }
            if (launch_pass(ctx, p, true)) return -1;
        }
    }
    return 0;
}

// extendPol restricted to the columns [first, first + ncols) of row-major sections of row stride ld (in: n rows, out: n_ext
// rows): the unit of the band-pipelined commit (xlp_dev_commit_from_host) -- a section is uploaded, extended and absorbed
// into its leaf hashes band by band.  Sizes outside the three-kernel fast path go through a dense copy of the band.
int dev_extend_cols(Ctx* ctx, u64* out, const u64* in, u64 n_ext, u64 n, u64 ld, u64 first, u64 ncols)
{
    if (n == 0 || ncols == 0) return 0;
    if (first + ncols > ld) return xlp_fail("extendPol (column window): window exceeds the row stride");
    if ((n & (n - 1)) || (n_ext & (n_ext - 1)) || n_ext < n) return xlp_fail("extendPol: sizes must be powers of two, n_ext >= n");
    u32 L = 0, Le = 0;
    while ((1ull << L) < n) L++;
    while ((1ull << Le) < n_ext) Le++;
    if (Le > 32) return xlp_fail("extendPol: size exceeds 2^32");
    if (lde_fast_supported(L, Le - L, ncols)) return dev_extend_fast_cols(ctx, out, in, L, Le - L, ld, first, ncols);
    if (first == 0 && ncols == ld) return dev_extend(ctx, out, in, n_ext, n, ncols, nullptr);
    u64* dense = nullptr;
    if (ctx_scratch(ctx, (n + n_ext) * ncols * sizeof(u64), 2, &dense)) return -1;
    if (dev_copy_cols(ctx, dense, ncols, in + first, ld, n, (u32)ncols)) return -1;
    if (dev_extend(ctx, dense + n * ncols, dense, n_ext, n, ncols, nullptr)) return -1;
    return dev_copy_cols(ctx, out + first, ld, dense + n * ncols, ncols, n_ext, (u32)ncols);
}

}  // namespace xlp
