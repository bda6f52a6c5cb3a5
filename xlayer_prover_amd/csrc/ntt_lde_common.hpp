// Shared declarations and the radix-16 round of the extendPol fast-path kernels (ntt_lde.hip, ntt_lde_mid.hip).
#pragma once
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "ntt_common.hpp"

template <int BIT> __device__ __forceinline__ gl::u64 gld(const gl::u64* q) { return *q; }
template <int BIT> __device__ __forceinline__ void gst(gl::u64* q, gl::u64 v) { *q = v; }
#define XLP_TWMUL(a, b) gl::mulc((a), (b))
#define XLP_SYNC() __syncthreads()

#ifndef XLP_WGT
#define XLP_WGT 1024     // threads per workgroup of the LDE kernels (experiment knob)
#endif

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
    const u64* tw2[LDE_MAX_COSETS];  // tw2[r][pos]: twiddle of the 2nd DIT round of lde_mid with the coset scale folded in
    const u64* hk[LDE_MAX_COSETS];   // hk[r][j] = (g_r^(2^BT))^(256 j), j < 16
    const u64* ur[LDE_MAX_COSETS];   // ur[r][j]   = g_r^j / N,       j  < 2^BT
};

// Scratch element (row, col) of a column chunk: the chunk's columns in groups of 8, every group a contiguous [2^L][8] matrix -- whatever the
// chunk's width (64 in a proof, 8 in the bench), a top-pass workgroup's 8 columns of a row are one 64-byte line and the middle pass's two
// 4-column workgroups share the lines of one group, exactly as in the 8-column bench (with [row][chunk width] rows of 512 bytes the middle
// pass ran 21 % slower inside a proof than in the bench: profiles/r3/lde_scratch_layout_ab.txt).
__host__ __device__ __forceinline__ u64 lde_sidx(u64 row, u32 col, u32 L) { return ((((u64)(col >> 3)) << L) + row) * 8 + (col & 7); }
// Scratch ROW of position (t, lo), t = the top BT bits: the rows of 8 neighbouring lo stay together, then t, then the rest of lo.  A top-pass workgroup (one lo,
// every t) then touches 64-byte pieces 512 bytes apart -- with its neighbours (lo +- 1, co-resident under xcd_tile) whole 512-byte runs -- instead of pieces
// 256 KiB apart; a middle-pass workgroup (one t, every lo) touches 512-byte runs 2^(BT+3) rows apart instead of one 256 KiB run.
#ifndef XLP_SROW_BITS
#define XLP_SROW_BITS 3       // measured at 2^23 x 8 on one box, ms per extendPol: rows as they come 2.31-2.37, 2 bits 2.36, 3 bits 2.26-2.29, 5 bits 2.30; only A or only B
#endif                        // re-indexed 2.29-2.34 / 2.28-2.31; blocked forms that also split t: middle pass 1.15 ms (profiles/r5/lde_scratch_rows_ab.txt)
__host__ __device__ __forceinline__ u64 lde_srow(u64 t, u32 lo, u32 BT)
{
    return ((u64)(lo >> XLP_SROW_BITS) << (BT + XLP_SROW_BITS)) | (t << XLP_SROW_BITS) | (lo & ((1u << XLP_SROW_BITS) - 1));
}
__host__ __device__ __forceinline__ u64 lde_swords(u32 wc, u32 L) { return ((u64)((wc + 7) >> 3) << L) * 8; }   // words of one scratch matrix of wc columns

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

constexpr u32 LDE_PF_LDS = 256;                              // bytes of dummy LDS in front of the tile (one wave instruction = 64 x 4 B)
// M0 (reserved by the compiler) is saved and restored around the load; one wait state between its write and the LDS-direct load.
__device__ __forceinline__ void prefetch_touch(const u64* q, u32 lds_off)
{
    u32 m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "v"(q), "s"(lds_off) : "memory");
}

// RESIDENT forms of the three extendPol kernels (xlp_dev_lde_resident, the ALU ceiling of bench.py's headline, SURVEY.md 8d: "measured by a
// register-resident microbenchmark"): the same kernels -- butterfly networks, twiddle tables, LDS tiles, barriers, grid -- with the global DATA traffic
// removed: an input word is made from its index (res_word), an output word is folded into a per-lane checksum that is stored only if it takes
// an impossible value (res_sink).  What a launch of the resident form takes is what the arithmetic and the LDS exchanges cost by themselves.
__device__ __forceinline__ u64 res_word(u64 index) { return index ^ 0x5EED0001ULL; }                       // (< 2^63: canonical; the kernels' timing does not depend on values)
__device__ __forceinline__ void res_sink(u64* out, u64 sink) { if (sink == 0x5EED5EED5EED5EEDULL) out[threadIdx.x] = sink; }

struct NoFactor { __device__ __forceinline__ u64 operator()(u32, u32) const { return 1; } };   // (tile position, column of the tile)


__device__ __forceinline__ u64 two_level12(const u64* __restrict__ lo, const u64* __restrict__ hi, u64 e)
{
    const u64 t = lo[e & 4095];
    const u64 eh = e >> 12;
    return eh ? gl::mul(t, hi[eh]) : t;
}


// One radix-2^D round over the tile digit at bit offset TP (compile-time geometry).
// Written in explicit phases -- (1) all twiddle loads, (2) all data loads, (3) arithmetic,
// (4) all stores -- because hipcc otherwise serialises "load; s_waitcnt vmcnt(0); use" per
// element (a full L2 round trip each).  pre/post return per-position factors (LDS cross
// twiddles) and are only evaluated when HAS_PRE / HAS_POST.
// VHI: the tile's column is the HIGH part of the group index (v = g >> (B - D): wave-uniform, adjacent lanes = adjacent positions) instead
// of the low part (adjacent lanes = adjacent columns): the column-major kernels' middle pass, whose columns are far apart in memory.
template <int B, int V, int TP, int D, bool DIT, bool INV, bool HAS_PRE, bool HAS_POST, class Load, class Store, class Pre, class Post, bool POSTW = false, bool VHI = false>
__device__ __forceinline__ void fast_round(const u64* __restrict__ tw, Load load, Store store, Pre pre, Post post)
{
    constexpr int RAD = 1 << D;
    constexpr u32 VB = V == 8 ? 3 : V == 4 ? 2 : V == 2 ? 1 : 0;
    constexpr u32 NG = (1u << (B - D)) * V;
    constexpr u32 TSHIFT = B - TP - D;
    for (u32 g = threadIdx.x; g < NG; g += blockDim.x) {
        const u32 v = VHI ? g >> (B - D) : g & (V - 1);
        const u32 rest = VHI ? g & ((1u << (B - D)) - 1) : g >> VB;
        const u32 low = rest & ((1u << TP) - 1);
        const u32 high = rest >> TP;
        const u32 base = (high << (TP + D)) | low;
        // LDS row of input/output j is affine in j (no carries between the digit fields):
        // lds_row(base + (j << TP)) = lds_row(base) + j * RS  -> one address register + immediates
        constexpr u32 RS = TP >= 4 ? (1u << TP) + ((1u << TP) >> 4) : (1u << TP);
        static_assert(TP == 0 || TP >= 4, "digit offsets are multiples of 4");
        const u32 r0 = lds_row(base);
        u64 twr[RAD];
        if constexpr (POSTW) {
            // position-indexed twiddles (scale folded in): one factor per input, j = 0 included
#pragma unroll
            for (int j = 0; j < RAD; j++) twr[j] = tw[base + ((u32)j << TP)];
        } else if constexpr (TP > 0) {
#pragma unroll
            for (int j = 1; j < RAD; j++) {
                twr[j] = tw[((u32)j * low) << TSHIFT];
            }
        }
        u64 x[RAD];
#pragma unroll
        for (int j = 0; j < RAD; j++) x[j] = load(base + ((u32)j << TP), v, r0 + (u32)j * RS);
        if constexpr (HAS_PRE) {
            u64 f[RAD];
#pragma unroll
            for (int j = 0; j < RAD; j++) f[j] = pre(base + ((u32)j << TP), v);
#pragma unroll
            for (int j = 0; j < RAD; j++) x[j] = XLP_TWMUL(x[j], f[j]);
        }
        if constexpr (POSTW) {
#pragma unroll
            for (int j = 0; j < RAD; j++) x[j] = XLP_TWMUL(x[j], twr[j]);
        } else if constexpr (DIT && TP > 0) {
#pragma unroll
            for (int j = 1; j < RAD; j++) x[j] = XLP_TWMUL(x[j], twr[j]);
        }
        dft_regs<D>(x);
        u64 y[RAD];
#pragma unroll
        for (int k = 0; k < RAD; k++) y[k] = x[brev<D>(INV ? (RAD - k) % RAD : k)];
        if constexpr (!DIT && TP > 0) {
#pragma unroll
            for (int k = 1; k < RAD; k++) y[k] = XLP_TWMUL(y[k], twr[k]);
        }
        if constexpr (HAS_POST) {
            u64 f[RAD];
#pragma unroll
            for (int k = 0; k < RAD; k++) f[k] = post(base + ((u32)k << TP), v);
#pragma unroll
            for (int k = 0; k < RAD; k++) y[k] = XLP_TWMUL(y[k], f[k]);
        }
#pragma unroll
        for (int k = 0; k < RAD; k++) store(base + ((u32)k << TP), v, y[k], r0 + (u32)k * RS);
    }
}
template <int B, int V, int TP, int D, bool DIT, bool INV, class Load, class Store>
__device__ __forceinline__ void fast_round(const u64* __restrict__ tw, Load load, Store store)
{
    fast_round<B, V, TP, D, DIT, INV, false, false>(tw, load, store, NoFactor{}, NoFactor{});
}


}  // namespace xlp
