// Shared declarations of the extendPol fast-path kernels (ntt_lde.hip).
#pragma once
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
    const u64* tw2[LDE_MAX_COSETS];  // tw2[r][pos]: twiddle of the 2nd DIT round of lde_mid with the coset scale folded in
    const u64* hk[LDE_MAX_COSETS];   // hk[r][j] = (g_r^(2^BT))^(256 j), j < 16
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

struct NoFactor { __device__ __forceinline__ u64 operator()(u32) const { return 1; } };


__device__ __forceinline__ u64 two_level12(const u64* __restrict__ lo, const u64* __restrict__ hi, u64 e)
{
    const u64 t = lo[e & 4095];
    const u64 eh = e >> 12;
    return eh ? gl::mul(t, hi[eh]) : t;
}


}  // namespace xlp
