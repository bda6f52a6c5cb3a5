// In-register radix-2^D Goldilocks DFT shared by the NTT kernels.
#pragma once
#include "gl_field.hpp"

namespace xlp {
using gl::u32;
using gl::u64;

// ------------------------------------------------- in-register 2^D-point DFT
// Forward DFT with w = w(D) = 2^(192 >> D) (w(4)=2^12, w(3)=2^24, w(2)=2^48,
// w(1)=-1): DIF network, x[brev(k)] = X[k] on return.  Canonical in and out.
template <int D>
__device__ __forceinline__ constexpr int brev(int k)
{
    int r = 0;
    for (int i = 0; i < D; i++) r |= ((k >> i) & 1) << (D - 1 - i);
    return r;
}

template <int D, int T, int BLK, int J>
__device__ __forceinline__ void bfly(u64 (&x)[1 << D])
{
    constexpr int half = 1 << (D - 1 - T);
    constexpr int i0 = BLK * 2 * half + J, i1 = i0 + half;
    constexpr int sh = J * (96 / half);  // w_{2half}^J = 2^(J*96/half)
    const u64 a = x[i0], b = x[i1];
    x[i0] = gl::addc(a, b);
    const u64 d = gl::subc(a, b);
    if constexpr (sh == 0) x[i1] = d;
    else x[i1] = gl::mul_pow2c<sh>(d);   // canonical
}
template <int D, int T, int IDX>
__device__ __forceinline__ void stage_rec(u64 (&x)[1 << D])
{
    constexpr int half = 1 << (D - 1 - T);
    if constexpr (IDX < (1 << (D - 1))) {
        bfly<D, T, IDX / half, IDX % half>(x);
        stage_rec<D, T, IDX + 1>(x);
    }
}
template <int D, int T>
__device__ __forceinline__ void dft_rec(u64 (&x)[1 << D])
{
    if constexpr (T < D) {
        stage_rec<D, T, 0>(x);
        dft_rec<D, T + 1>(x);
    }
}
template <int D>
__device__ __forceinline__ void dft_regs(u64 (&x)[1 << D])
{
    dft_rec<D, 0>(x);
}


__device__ __forceinline__ u32 lds_row(u32 pos) { return pos + (pos >> 4); }   // 1 pad row per 16: conflict-free stride-16 access

}  // namespace xlp
