// Goldilocks NTT / INTT / LDE (extendPol) for gfx950.
//
// Replaces NTT_Goldilocks::NTT / INTT / extendPol of the reference's absent
// goldilocks submodule as called from src/starkpil/starks.cpp:53,134,215
// (extendPol), :262,285,326-327 (NTT/INTT) and friProve.cpp:100-102.
//
// Design (MI355X-first, not a translation of the AVX code):
//  * data stays row-major [rows][ncols] as in the reference slab
//    (stark_info.cpp:473-482); a workgroup owns a TILE = 2^b points of one
//    strided sub-sequence x 8 "vector lanes" (8 adjacent columns, or adjacent
//    rows x columns when ncols < 8) so every global access is a 64-byte
//    segment, and tiles that are adjacent in memory are given to workgroups
//    that run on the same XCD at the same time (shared L2 merges the lines).
//  * a size-2^L transform is ceil(L/11) passes; inside a pass the tile sits in
//    LDS (<= 136 KiB) and is processed by radix-16 butterflies held in VGPRs:
//    the 64th roots of unity of Goldilocks are powers of two (w(6) = 8), so a
//    16-point DFT needs only shifts/adds; one general twiddle multiply per
//    element per radix-16 round.
//  * INTT runs decimation-in-frequency (natural in, digit-reversed out) and
//    the forward NTT of the LDE runs decimation-in-time (digit-reversed in,
//    natural out), so no bit-reversal pass ever touches HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gl_field.hpp"
#include "xlp_internal.hpp"

namespace xlp {
using gl::u32;
using gl::u64;

// ------------------------------------------------------------------ params
constexpr int MAX_DIGITS = 12;

enum RowMode : u32 {
    ROW_POS = 0,      // row = position
    ROW_NATURAL = 1,  // row = digit-reversed position (natural transform index)
    ROW_COSET = 2,    // row = (position << ext_bits) | coset   (LDE output interleave)
};

struct PassParams {
    const u64* src;
    u64* dst;
    u64 ncols;         // row stride (elements) of src and dst
    u32 L;             // log2 of transform size
    u32 s, b;          // this pass transforms position bits [s, s+b)
    u32 C, R;          // vector lane v in [0,8): r = v / C (adjacent low rows), c = v % C (columns); valid if r < R
    u32 n_colchunks;   // ceil(ncols / C)
    u32 n_logroups;    // ceil(2^s / R)
    u64 ntiles;        // real tile count (grid is padded to a multiple of 8)
    u32 inverse;       // use inverse 16-point DFT ordering (tables hold inverse roots)
    u32 dit;           // 0: DIF (butterfly then twiddle), 1: DIT (twiddle then butterfly)
    u32 src_mode, dst_mode;
    u32 ext_bits, coset;
    u32 n_digits;      // global digit list (top -> bottom) for digit reversal
    u32 digit_bits[MAX_DIGITS];
    const u64* tw_lo;  // tw_lo[i] = w^i, i < 4096            (w = w(L) or its inverse)
    const u64* tw_hi;  // tw_hi[i] = w^(i << 12), i < 2^(L-12)   (L > 12 only)
    const u64* sc_lo;  // optional load-scale tables, indexed by the natural (coefficient) index
    const u64* sc_hi;
    u64 final_scale;   // multiplied into every stored element if != 1 (1/n of a stand-alone INTT)
};

__device__ __forceinline__ u64 digit_reverse(const PassParams& p, u64 pos)
{
    u64 k = 0;
    u32 consumed = 0, shift = 0;
    for (u32 i = 0; i < p.n_digits; i++) {
        u32 d = p.digit_bits[i];
        consumed += d;
        u64 dv = (pos >> (p.L - consumed)) & ((1u << d) - 1);
        k |= dv << shift;
        shift += d;
    }
    return k;
}

__device__ __forceinline__ u64 twiddle(const PassParams& p, u64 e)
{
    // w^e, e < 2^L
    u64 t = p.tw_lo[e & 4095];
    if (p.L > 12) {
        u64 eh = e >> 12;
        if (eh) t = gl::mul(t, p.tw_hi[eh]);
    }
    return t;
}

// ------------------------------------------------- in-register 2^D-point DFT
// Forward DFT with w = w(D) = 2^(192 >> D) (w(4)=2^12, w(3)=2^24, w(2)=2^48,
// w(1)=-1): DIF network, x[bitrev(k)] = X[k] on return.  Values canonical.
template <int D>
__device__ __forceinline__ constexpr int brev(int k)
{
    int r = 0;
    for (int i = 0; i < D; i++) r |= ((k >> i) & 1) << (D - 1 - i);
    return r;
}

template <int D>
__device__ __forceinline__ void dft_regs(u64 (&x)[1 << D])
{
#pragma unroll
    for (int t = 0; t < D; t++) {
        const int half = 1 << (D - 1 - t);
#pragma unroll
        for (int blk = 0; blk < (1 << t); blk++) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                const int i0 = blk * 2 * half + j, i1 = i0 + half;
                u64 a = x[i0], b = x[i1];
                x[i0] = gl::addc(a, b);
                u64 d = gl::subc(a, b);
                const int sh = j * (96 / half);  // w_{2half}^j = 2^(j*96/half)
                x[i1] = sh ? gl::canon(gl::mul_pow2(d, sh)) : d;
            }
        }
    }
}

// --------------------------------------------------------------- tile engine
struct TileCtx {
    const PassParams& p;
    u64* sm;           // LDS tile: row(pos) * 8 + v
    u64 hi, lo0;       // tile coordinates
    u32 c0;
};

__device__ __forceinline__ u32 lds_row(u32 pos) { return pos + (pos >> 4); }

__device__ __forceinline__ bool lane_valid(const TileCtx& t, u32 v, u32& r, u32& c)
{
    r = v / t.p.C;
    c = v % t.p.C;
    return r < t.p.R && (t.lo0 + r) < (1ull << t.p.s) && (t.c0 + c) < t.p.ncols;
}

__device__ __forceinline__ u64 global_pos(const TileCtx& t, u32 tpos, u32 r)
{
    return (t.hi << (t.p.s + t.p.b)) | ((u64)tpos << t.p.s) | (t.lo0 + r);
}

__device__ __forceinline__ u64 map_row(const PassParams& p, u32 mode, u64 pos)
{
    if (mode == ROW_POS) return pos;
    if (mode == ROW_NATURAL) return digit_reverse(p, pos);
    return (pos << p.ext_bits) | p.coset;
}

// One radix-2^D round over tile digit at bit offset tp.  FIRST: inputs come
// from global memory; LAST: outputs go to global memory; otherwise LDS.
template <int D>
__device__ __forceinline__ void round_body(const TileCtx& t, u32 tp, bool first, bool last)
{
    const PassParams& p = t.p;
    constexpr int RAD = 1 << D;
    const u32 ngroups = (1u << (p.b - D)) * 8;
    const u32 gpos_bits = p.s + tp;  // global bit offset of this digit
    for (u32 g = threadIdx.x; g < ngroups; g += blockDim.x) {
        const u32 v = g & 7;
        const u32 rest = g >> 3;
        const u32 low = rest & ((1u << tp) - 1);
        const u32 high = rest >> tp;
        const u32 base = (high << (tp + D)) | low;
        u32 r, c;
        const bool valid = lane_valid(t, v, r, c);
        // twiddle exponent factor: (global low bits below the digit) << (L - gpos_bits - D)
        const u64 lowglobal = valid ? (((u64)low << p.s) | (t.lo0 + r)) : 0;  // invalid lanes: exponent 0
        const u32 tshift = p.L - gpos_bits - D;
        const bool has_tw = gpos_bits > 0;
        u64 x[RAD];
        // ---- gather (+ DIT twiddle)
#pragma unroll
        for (int j = 0; j < RAD; j++) {
            const u32 tpos = base + ((u32)j << tp);
            u64 val = 0;
            if (first) {
                if (valid) {
                    const u64 pos = global_pos(t, tpos, r);
                    const u64 row = map_row(p, p.src_mode, pos);
                    val = gl::canon(p.src[row * p.ncols + t.c0 + c]);
                    if (p.sc_lo) {
                        const u64 ci = digit_reverse(p, pos);  // coefficient index
                        u64 sc = p.sc_lo[ci & 4095];
                        if (p.L > 12) sc = gl::mul(sc, p.sc_hi[ci >> 12]);
                        val = gl::mulc(val, sc);
                    }
                }
            } else {
                val = t.sm[lds_row(tpos) * 8 + v];
            }
            if (p.dit && has_tw && j) val = gl::mulc(val, twiddle(p, ((u64)j * lowglobal) << tshift));
            x[j] = val;
        }
        dft_regs<D>(x);
        // ---- scatter (+ DIF twiddle)
#pragma unroll
        for (int k = 0; k < RAD; k++) {
            // forward: X[k] = x[brev(k)];  inverse: X^-1[k] = X[(RAD-k)%RAD]
            u64 val = p.inverse ? x[brev<D>((RAD - k) % RAD)] : x[brev<D>(k)];
            if (!p.dit && has_tw && k) val = gl::mulc(val, twiddle(p, ((u64)k * lowglobal) << tshift));
            const u32 tpos = base + ((u32)k << tp);
            if (last) {
                if (valid) {
                    if (p.final_scale != 1) val = gl::mulc(val, p.final_scale);
                    const u64 pos = global_pos(t, tpos, r);
                    const u64 row = map_row(p, p.dst_mode, pos);
                    p.dst[row * p.ncols + t.c0 + c] = val;
                }
            } else {
                t.sm[lds_row(tpos) * 8 + v] = val;
            }
        }
    }
}

__device__ __forceinline__ void round_dispatch(const TileCtx& t, u32 d, u32 tp, bool first, bool last)
{
    switch (d) {
        case 1: round_body<1>(t, tp, first, last); break;
        case 2: round_body<2>(t, tp, first, last); break;
        case 3: round_body<3>(t, tp, first, last); break;
        default: round_body<4>(t, tp, first, last); break;
    }
}

__global__ __launch_bounds__(1024) void ntt_pass_kernel(const PassParams p)
{
    extern __shared__ u64 sm[];
    // ---- XCD-aware tile assignment: blocks b, b+8, b+16.. share an XCD and
    // run concurrently; give them tiles that are adjacent in memory
    // (column-chunk fastest, then low-row group, then hi).
    const u64 bid = blockIdx.x;
    u64 tile = bid;
    if (gridDim.x >= 16) tile = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);  // gridDim.x % 8 == 0 here
    if (tile >= p.ntiles) return;  // block-uniform: grid padding only
    TileCtx t{p, sm, 0, 0, 0};
    const u64 cc = tile % p.n_colchunks;
    const u64 rest = tile / p.n_colchunks;
    t.c0 = (u32)(cc * p.C);
    t.lo0 = (rest % p.n_logroups) * p.R;
    t.hi = rest / p.n_logroups;

    const u32 m = (p.b + 3) / 4;
    const u32 dtop = p.b - 4 * (m - 1);
    if (!p.dit) {
        for (u32 r = 0; r < m; r++) {
            const u32 d = r == 0 ? dtop : 4;
            const u32 tp = 4 * (m - 1 - r);
            round_dispatch(t, d, tp, r == 0, r == m - 1);
            if (r + 1 < m) __syncthreads();
        }
    } else {
        for (u32 q = 0; q < m; q++) {
            const u32 r = m - 1 - q;
            const u32 d = r == 0 ? dtop : 4;
            const u32 tp = 4 * (m - 1 - r);
            round_dispatch(t, d, tp, q == 0, q == m - 1);
            if (q + 1 < m) __syncthreads();
        }
    }
}

// -------------------------------------------------------------- table kernels
// out[i] = base^(i << shift), i < count
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

// Power tables of `root` over exponents < 2^L, split 12/L-12 (device memory, cached in ctx).
static int get_tables(Ctx* ctx, u32 L, u64 root, u64 mult, const u64** lo, const u64** hi)
{
    auto key = std::make_tuple(L, root, mult);
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) {
        u64 nlo = L < 12 ? (1ull << L) : 4096, nhi = L > 12 ? (1ull << (L - 12)) : 0;
        u64* d = nullptr;
        XLP_HIP(hipMalloc(&d, (4096 + nhi) * sizeof(u64)));
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
    u32 b[4], s[4];
    u32 n_digits;
    u32 digit_bits[MAX_DIGITS];
};

static Plan make_plan(u32 L)
{
    Plan pl{};
    pl.L = L;
    pl.np = L == 0 ? 0 : (L + 10) / 11;
    u32 acc = 0;
    for (u32 i = 0; i < pl.np; i++) pl.b[i] = L / pl.np + (i < L % pl.np ? 1 : 0);
    for (u32 i = pl.np; i-- > 0;) { pl.s[i] = acc; acc += pl.b[i]; }
    pl.n_digits = 0;
    for (u32 i = 0; i < pl.np; i++) {
        u32 m = (pl.b[i] + 3) / 4, dtop = pl.b[i] - 4 * (m - 1);
        for (u32 r = 0; r < m; r++) pl.digit_bits[pl.n_digits++] = r == 0 ? dtop : 4;
    }
    return pl;
}

static int launch_pass(Ctx* ctx, PassParams& p)
{
    const u64 ncols = p.ncols;
    p.C = ncols < 8 ? (u32)ncols : 8;
    u32 R = 1;
    while (R * 2 * p.C <= 8 && (u64)R * 2 <= (1ull << p.s)) R *= 2;
    p.R = R;
    p.n_colchunks = (u32)((ncols + p.C - 1) / p.C);
    p.n_logroups = (u32)(((1ull << p.s) + R - 1) / R);
    u64 ntiles = (u64)p.n_colchunks * p.n_logroups * (1ull << (p.L - p.s - p.b));
    u64 grid = ntiles < 16 ? ntiles : (ntiles + 7) / 8 * 8;
    if (grid > 0x7fffffffULL) return xlp_fail("ntt: grid too large");
    u32 groups = (1u << (p.b > 4 ? p.b - 4 : 0)) * 8;
    u32 threads = groups < 64 ? 64 : (groups > 1024 ? 1024 : groups);
    u32 rows = 1u << p.b;
    size_t lds = (size_t)(rows + (rows >> 4) + 1) * 8 * sizeof(u64);
    p.ntiles = ntiles;
    hipLaunchKernelGGL(ntt_pass_kernel, dim3((unsigned)grid), dim3(threads), lds, ctx->stream, p);
    XLP_HIP(hipGetLastError());
    return 0;
}

static void fill_digits(PassParams& p, const Plan& pl)
{
    p.n_digits = pl.n_digits;
    for (u32 i = 0; i < pl.n_digits; i++) p.digit_bits[i] = pl.digit_bits[i];
}

int ntt_init(Ctx* ctx)
{
    XLP_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    (void)ctx;
    return 0;
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
    Plan pl = make_plan(L);
    u64 root = host_w(L);
    if (inverse) root = hgl::inv(root);
    const u64 *tlo, *thi;
    if (get_tables(ctx, L, root, 1, &tlo, &thi)) return -1;
    u64* work = nullptr;
    if (pl.np > 1) {
        if (ctx_scratch(ctx, n * ncols * sizeof(u64), 0, &work)) return -1;
    }
    for (u32 i = 0; i < pl.np; i++) {
        PassParams p{};
        p.ncols = ncols; p.L = L; p.s = pl.s[i]; p.b = pl.b[i];
        p.inverse = inverse ? 1 : 0; p.dit = 0;
        p.tw_lo = tlo; p.tw_hi = thi; p.sc_lo = nullptr; p.sc_hi = nullptr;
        p.final_scale = 1;
        fill_digits(p, pl);
        const bool first = i == 0, last = i == pl.np - 1;
        p.src = first ? src : work;
        p.dst = last ? dst : work;
        p.src_mode = ROW_POS;
        p.dst_mode = last ? ROW_NATURAL : ROW_POS;
        if (last && inverse) p.final_scale = hgl::inv(n % hgl::P);
        if (launch_pass(ctx, p)) return -1;
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
    if (get_tables(ctx, L, hgl::inv(wn), 1, &ilo, &ihi)) return -1;
    if (get_tables(ctx, L, wn, 1, &flo, &fhi)) return -1;
    for (u32 i = 0; i < pl.np; i++) {
        PassParams p{};
        p.ncols = ncols; p.L = L; p.s = pl.s[i]; p.b = pl.b[i];
        p.inverse = 1; p.dit = 0;
        p.tw_lo = ilo; p.tw_hi = ihi;
        p.final_scale = 1;
        fill_digits(p, pl);
        p.src = i == 0 ? in : A;
        p.dst = A;
        p.src_mode = ROW_POS; p.dst_mode = ROW_POS;
        if (launch_pass(ctx, p)) return -1;
    }
    // ---- per coset r: coefficients * (shift*w_ext^r)^i / n, then DIT NTT -> out rows (q << e) | r
    const u64 wext = host_w(Le);
    for (u64 r = 0; r < ncoset; r++) {
        const u64 g = hgl::mul(7, hgl::pow(wext, r));
        const u64 *slo, *shi;
        if (get_tables(ctx, L, g, ninv, &slo, &shi)) return -1;
        for (u32 q = 0; q < pl.np; q++) {
            const u32 i = pl.np - 1 - q;  // DIT: bottom pass first
            PassParams p{};
            p.ncols = ncols; p.L = L; p.s = pl.s[i]; p.b = pl.b[i];
            p.inverse = 0; p.dit = 1;
            p.tw_lo = flo; p.tw_hi = fhi;
            p.final_scale = 1;
            fill_digits(p, pl);
            const bool first = q == 0, last = q == pl.np - 1;
            p.src = first ? A : B;
            p.dst = last ? out : B;
            p.src_mode = ROW_POS;
            p.dst_mode = last ? ROW_COSET : ROW_POS;
            p.ext_bits = e; p.coset = (u32)r;
            if (first) { p.sc_lo = slo; p.sc_hi = shi; }
            if (launch_pass(ctx, p)) return -1;
        }
    }
    return 0;
}

}  // namespace xlp
