// Row-scale helpers of Starks::genProof that sit between the commit stages:
//   Polinomial::batchInverse / batchInverseParallel   src/starkpil/polinomial.hpp:635-720  (starks.cpp:358-359)
//   Polinomial::calculateZ                            src/starkpil/polinomial.hpp:586-607  (starks.cpp:180-185)
//   Starks::evmap                                     src/starkpil/starks.cpp:556-669      (mulAddElement_adim3 polinomial.hpp:722-744)
// All are scan / reduce / map shapes over 2^23..2^24 rows of Fp or Fp^3 elements.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "../../include/xlp_gl.h"

namespace xlp {
using gl::e3;
using gl::u32;
using gl::u64;

__device__ __forceinline__ u64 inv1(u64 a)   // a^(p-2); inv(0) = 0
{
    // p - 2 = 0xFFFFFFFEFFFFFFFF : 32 ones, a zero, 31 ones... plain square-and-multiply (off the hot path)
    u64 r = 1, b = a;
    u64 e = gl::P - 2;
#pragma unroll 1
    for (int i = 0; i < 64; i++) {
        if (e & 1) r = gl::mul(r, b);
        b = gl::mul(b, b);
        e >>= 1;
    }
    return gl::canon(r);
}

// Inverse in Fp[x]/(x^3 - x - 1): first column of adj(M_a) / det(M_a), M_a = matrix of "multiply by a".
__device__ __forceinline__ e3 inv3(const e3& a)
{
    const u64 a0 = a.c[0], a1 = a.c[1], a2 = a.c[2];
    const u64 m11 = gl::addc(a0, a2), m12 = gl::addc(a1, a2), m22 = m11;
    const u64 c00 = gl::subc(gl::mulc(m11, m22), gl::mulc(m12, a1));
    const u64 c01 = gl::subc(gl::mulc(m12, a2), gl::mulc(a1, m22));
    const u64 c02 = gl::subc(gl::mulc(a1, a1), gl::mulc(m11, a2));
    const u64 det = gl::addc(gl::addc(gl::mulc(a0, c00), gl::mulc(a2, c01)), gl::mulc(a1, c02));
    const u64 di = inv1(det);
    e3 r;
    r.c[0] = gl::mulc(c00, di); r.c[1] = gl::mulc(c01, di); r.c[2] = gl::mulc(c02, di);
    return r;
}

__device__ __forceinline__ e3 load3(const u64* p)
{
    e3 r; r.c[0] = gl::canon(p[0]); r.c[1] = gl::canon(p[1]); r.c[2] = gl::canon(p[2]); return r;
}
__device__ __forceinline__ void store3(u64* p, const e3& v) { p[0] = v.c[0]; p[1] = v.c[1]; p[2] = v.c[2]; }
__device__ __forceinline__ e3 one3() { e3 r; r.c[0] = 1; r.c[1] = 0; r.c[2] = 0; return r; }
__device__ __forceinline__ bool is_zero3(const e3& a) { return (a.c[0] | a.c[1] | a.c[2]) == 0; }

// res[i] = 1 / src[i] (dim 3).  Each lane inverts CH consecutive elements with Montgomery's trick
// (one field inversion per CH elements); zero elements are passed through as zero.
constexpr int INV_CH = 4;
__global__ __launch_bounds__(256) void batch_inverse3_kernel(u64* res, const u64* src, u64 n)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 i0 = t * INV_CH;
    if (i0 >= n) return;
    e3 a[INV_CH], pre[INV_CH];
    bool z[INV_CH];
    e3 acc = one3();
#pragma unroll
    for (int k = 0; k < INV_CH; k++) {
        const bool ok = i0 + k < n;
        a[k] = ok ? load3(src + 3 * (i0 + k)) : one3();
        z[k] = is_zero3(a[k]);
        if (z[k]) a[k] = one3();
        pre[k] = acc;
        acc = gl::mul3(acc, a[k]);
    }
    e3 inv = inv3(acc);
#pragma unroll
    for (int k = INV_CH - 1; k >= 0; k--) {
        e3 r = gl::mul3(inv, pre[k]);
        inv = gl::mul3(inv, a[k]);
        if (z[k]) { r.c[0] = 0; r.c[1] = 0; r.c[2] = 0; }
        if (i0 + k < n) store3(res + 3 * (i0 + k), r);
    }
}

__global__ __launch_bounds__(256) void batch_inverse1_kernel(u64* res, const u64* src, u64 n)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) res[i] = inv1(gl::canon(src[i]));
}

// ------------------------------------------------------------- calculateZ = exclusive product scan
// r[i] = num[i] / den[i];  z[0] = 1, z[i] = prod_{j<i} r[j].   Three phases: per-block inclusive scan of
// SCAN_PER_BLOCK ratios (+ block product), recursive scan of the block products, apply.
constexpr int SCAN_T = 256, SCAN_E = 8, SCAN_PER_BLOCK = SCAN_T * SCAN_E;

// mode 0: v[i] = num[i] / den[i] ; mode 1: v[i] = num[i] (already ratios / block products)
__global__ __launch_bounds__(SCAN_T) void scan3_block_kernel(u64* out_incl, u64* block_prod, const u64* num, u64 ns, const u64* den, u64 ds, u64 n, int mode)
{
    __shared__ u64 sh[SCAN_T][3];
    const u64 base = (u64)blockIdx.x * SCAN_PER_BLOCK + (u64)threadIdx.x * SCAN_E;
    e3 v[SCAN_E];
    e3 acc = one3();
    // mode 0: the ratios num / den, the denominators inverted four at a time with Montgomery's trick (one field inversion = ~140 base
    // multiplications per FOUR elements instead of per element; a zero denominator gives a zero ratio, as batch_inverse3_kernel does)
    constexpr int ZB = 4;
#pragma unroll
    for (int k0 = 0; k0 < SCAN_E; k0 += ZB) {
        e3 x[ZB];
        if (mode == 0) {
            e3 d[ZB], pre[ZB];
            bool z[ZB];
            e3 pacc = one3();
#pragma unroll
            for (int k = 0; k < ZB; k++) {
                const u64 i = base + k0 + k;
                d[k] = i < n ? load3(den + ds * i) : one3();
                z[k] = is_zero3(d[k]);
                if (z[k]) d[k] = one3();
                pre[k] = pacc;
                pacc = gl::mul3(pacc, d[k]);
            }
            e3 inv = inv3(pacc);
#pragma unroll
            for (int k = ZB - 1; k >= 0; k--) {
                const u64 i = base + k0 + k;
                e3 r = gl::mul3(inv, pre[k]);
                inv = gl::mul3(inv, d[k]);
                if (z[k]) { r.c[0] = 0; r.c[1] = 0; r.c[2] = 0; }
                x[k] = i < n ? gl::mul3(load3(num + ns * i), r) : one3();
            }
        } else {
#pragma unroll
            for (int k = 0; k < ZB; k++) {
                const u64 i = base + k0 + k;
                x[k] = i < n ? load3(num + ns * i) : one3();
            }
        }
#pragma unroll
        for (int k = 0; k < ZB; k++) {
            acc = gl::mul3(acc, x[k]);
            v[k0 + k] = acc;             // inclusive within the thread
        }
    }
    sh[threadIdx.x][0] = acc.c[0]; sh[threadIdx.x][1] = acc.c[1]; sh[threadIdx.x][2] = acc.c[2];
    __syncthreads();
    // Hillis-Steele inclusive scan of the thread totals
    for (int off = 1; off < SCAN_T; off <<= 1) {
        e3 mine, other;
        const bool act = (int)threadIdx.x >= off;
        mine.c[0] = sh[threadIdx.x][0]; mine.c[1] = sh[threadIdx.x][1]; mine.c[2] = sh[threadIdx.x][2];
        if (act) { other.c[0] = sh[threadIdx.x - off][0]; other.c[1] = sh[threadIdx.x - off][1]; other.c[2] = sh[threadIdx.x - off][2]; }
        __syncthreads();
        if (act) {
            mine = gl::mul3(mine, other);
            sh[threadIdx.x][0] = mine.c[0]; sh[threadIdx.x][1] = mine.c[1]; sh[threadIdx.x][2] = mine.c[2];
        }
        __syncthreads();
    }
    e3 pre = one3();
    if (threadIdx.x > 0) { pre.c[0] = sh[threadIdx.x - 1][0]; pre.c[1] = sh[threadIdx.x - 1][1]; pre.c[2] = sh[threadIdx.x - 1][2]; }
#pragma unroll
    for (int k = 0; k < SCAN_E; k++) {
        const u64 i = base + k;
        if (i < n) store3(out_incl + 3 * i, gl::mul3(pre, v[k]));
    }
    if (threadIdx.x == SCAN_T - 1 && block_prod) {
        block_prod[3 * blockIdx.x + 0] = sh[SCAN_T - 1][0];
        block_prod[3 * blockIdx.x + 1] = sh[SCAN_T - 1][1];
        block_prod[3 * blockIdx.x + 2] = sh[SCAN_T - 1][2];
    }
}

// incl[i] *= prefix of the blocks before block(i)   (block_incl = inclusive scan of block products)
__global__ __launch_bounds__(256) void scan3_apply_kernel(u64* incl, const u64* block_incl, u64 n)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 b = i / SCAN_PER_BLOCK;
    if (b == 0) return;
    store3(incl + 3 * i, gl::mul3(load3(incl + 3 * i), load3(block_incl + 3 * (b - 1))));
}

// z[0] = 1, z[i] = incl[i-1]
__global__ __launch_bounds__(256) void scan3_shift_kernel(u64* z, u64 zs, const u64* incl, u64 n)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0) { z[0] = 1; z[1] = 0; z[2] = 0; return; }
    z[zs * i] = incl[3 * (i - 1)]; z[zs * i + 1] = incl[3 * (i - 1) + 1]; z[zs * i + 2] = incl[3 * (i - 1) + 2];
}

static int scan3_inclusive(Ctx* ctx, u64* incl, const u64* num, u64 ns, const u64* den, u64 ds, u64 n, int mode, u64* scratch, u64 scratch_elems)
{
    const u64 nb = (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    if (3 * nb > scratch_elems && nb > 1) return xlp_fail("scan: scratch too small");
    u64* bp = nb > 1 ? scratch : nullptr;
    hipLaunchKernelGGL(scan3_block_kernel, dim3((unsigned)nb), dim3(SCAN_T), 0, ctx->stream, incl, bp, num, ns, den, ds, n, mode);
    XLP_HIP(hipGetLastError());
    if (nb > 1) {
        u64* bincl = scratch + 3 * nb;
        if (6 * nb > scratch_elems) return xlp_fail("scan: scratch too small");
        if (scan3_inclusive(ctx, bincl, bp, 3, nullptr, 3, nb, 1, scratch + 6 * nb, scratch_elems - 6 * nb)) return -1;
        hipLaunchKernelGGL(scan3_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, incl, bincl, n);
        XLP_HIP(hipGetLastError());
    }
    return 0;
}

// z, num, den: element i at p[i * stride .. + 3) -- the reference's Polinomial(address, degree, dim, offset) views of a row-major
// section (polinomial.hpp:24-40), so that a column of a section needs no transposed copy around the call (starks.cpp:406-555).
int dev_calculate_z(Ctx* ctx, u64* z, u64 zs, const u64* num, u64 ns, const u64* den, u64 ds, u64 n)
{
    if (!n) return 0;
    if (zs < 3 || ns < 3 || ds < 3) return xlp_fail("calculateZ: stride below the dimension");
    u64* tmp = nullptr;
    const u64 nb = (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    const u64 aux = 8 * (nb + 64) * 3;
    if (ctx_scratch(ctx, (3 * n + aux) * sizeof(u64), 2, &tmp)) return -1;
    if (scan3_inclusive(ctx, tmp, num, ns, den, ds, n, 0, tmp + 3 * n, aux)) return -1;
    hipLaunchKernelGGL(scan3_shift_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, z, zs, tmp, n);
    XLP_HIP(hipGetLastError());
    return 0;
}

// z[i] <- z[i] * s for the Fp^3 elements of a strided view (row-sharded calculateZ: a rank's local scan times the product of the ranks before it)
__global__ __launch_bounds__(256) void scale3_kernel(u64* z, u64 zs, u64 n, e3 s)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store3(z + i * zs, gl::mul3(load3(z + i * zs), s));
}
int dev_scale3(Ctx* ctx, u64* z, u64 zs, u64 n, const u64 s[3])
{
    if (!n) return 0;
    if (zs < 3) return xlp_fail("scale3: stride below the dimension");
    e3 k; k.c[0] = s[0] % hgl::P; k.c[1] = s[1] % hgl::P; k.c[2] = s[2] % hgl::P;
    hipLaunchKernelGGL(scale3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, z, zs, n, k);
    XLP_HIP(hipGetLastError());
    return 0;
}

int dev_batch_inverse(Ctx* ctx, u64* res, const u64* src, u64 n, u32 dim)
{
    if (!n) return 0;
    if (dim == 3) {
        const u64 thr = (n + INV_CH - 1) / INV_CH;
        hipLaunchKernelGGL(batch_inverse3_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, ctx->stream, res, src, n);
    } else if (dim == 1) {
        hipLaunchKernelGGL(batch_inverse1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, res, src, n);
    } else return xlp_fail("batchInverse: dim must be 1 or 3");
    XLP_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ genProof tables (starks.hpp:149-183, starks.cpp:318-362)
// out[k] = first * ratio^k, k < n: x_n / x_2ns / x (dim 1: ratio = w(nBits), first = 1 or shift) and LEv / LpEv before
// their INTT (dim 3: ratio = xi / shift).  The reference builds them with n sequential multiplications; here lane t
// starts from ratio^t (square and multiply) and strides by ratio^T, so the writes are coalesced.
__device__ __forceinline__ e3 pow3(e3 b, u64 e)
{
    e3 r = one3();
    while (e) { if (e & 1) r = gl::mul3(r, b); b = gl::mul3(b, b); e >>= 1; }
    return r;
}
__global__ __launch_bounds__(256) void geometric_kernel(u64* out, u64 n, u32 dim, e3 first, e3 ratio)
{
    const u64 T = (u64)gridDim.x * blockDim.x, t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (dim == 1) {
        u64 v = gl::mulc(first.c[0], gl::pow(ratio.c[0], t));
        const u64 step = gl::canon(gl::pow(ratio.c[0], T));
        for (u64 k = t; k < n; k += T) { out[k] = v; v = gl::mulc(v, step); }
    } else {
        e3 v = gl::mul3(first, pow3(ratio, t));
        const e3 step = pow3(ratio, T);
        for (u64 k = t; k < n; k += T) { store3(out + 3 * k, v); v = gl::mul3(v, step); }
    }
}

int dev_geometric(Ctx* ctx, u64* out, u64 n, u32 dim, const u64* first, const u64* ratio)
{
    if (!n) return 0;
    if (dim != 1 && dim != 3) return xlp_fail("geometric: dim must be 1 or 3");
    e3 f, r;
    for (int c = 0; c < 3; c++) { f.c[c] = c < (int)dim ? first[c] % hgl::P : 0; r.c[c] = c < (int)dim ? ratio[c] % hgl::P : 0; }
    u64 blocks = (n + 255) / 256;
    if (blocks > (u64)ctx->ncu * 8) blocks = (u64)ctx->ncu * 8;
    hipLaunchKernelGGL(geometric_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, out, n, dim, f, r);
    XLP_HIP(hipGetLastError());
    return 0;
}

// out[k] = x[k] / (x[k] - xi) in Fp^3, x[k] in Fp (starks.cpp:340-362: subElement, batchInverseParallel, mulElement),
// one pass: each lane inverts INV_CH differences with Montgomery's trick.  x[k] = xi (zero difference) yields zero,
// as xlp_dev_batch_inverse does.
__global__ __launch_bounds__(256) void x_div_x_sub_kernel(u64* out, const u64* x, e3 xi, u64 n)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 i0 = t * INV_CH;
    if (i0 >= n) return;
    e3 a[INV_CH], pre[INV_CH];
    u64 xv[INV_CH];
    bool z[INV_CH];
    e3 acc = one3();
#pragma unroll
    for (int k = 0; k < INV_CH; k++) {
        const bool ok = i0 + k < n;
        xv[k] = ok ? gl::canon(x[i0 + k]) : 0;
        a[k].c[0] = gl::subc(xv[k], xi.c[0]); a[k].c[1] = gl::neg(xi.c[1]); a[k].c[2] = gl::neg(xi.c[2]);
        if (!ok) a[k] = one3();
        z[k] = is_zero3(a[k]);
        if (z[k]) a[k] = one3();
        pre[k] = acc;
        acc = gl::mul3(acc, a[k]);
    }
    e3 inv = inv3(acc);
#pragma unroll
    for (int k = INV_CH - 1; k >= 0; k--) {
        e3 r = gl::mul3s(gl::mul3(inv, pre[k]), xv[k]);
        inv = gl::mul3(inv, a[k]);
        if (z[k]) { r.c[0] = 0; r.c[1] = 0; r.c[2] = 0; }
        if (i0 + k < n) store3(out + 3 * (i0 + k), r);
    }
}

int dev_x_div_x_sub_xi(Ctx* ctx, u64* out, const u64* x, const u64* xi, u64 n)
{
    if (!n) return 0;
    e3 v;
    for (int c = 0; c < 3; c++) v.c[c] = xi[c] % hgl::P;
    const u64 thr = (n + INV_CH - 1) / INV_CH;
    hipLaunchKernelGGL(x_div_x_sub_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, ctx->stream, out, x, v, n);
    XLP_HIP(hipGetLastError());
    return 0;
}

// Step-4 coefficient split (starks.cpp:267-283): qq2[k * qDeg + p] = qq1[p * N + k] * shiftIn^p for k < N, p < qDeg
// (elements of qDim words); rows N .. NExtended-1 of the (qDeg * qDim)-column matrix the following NTT reads are zero.
__global__ __launch_bounds__(256) void q_split_kernel(u64* qq2, const u64* qq1, u64 n, u64 n_ext, u32 qdeg, u32 qdim, u64 shift_in)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;      // one (row, p) pair
    if (t >= n_ext * qdeg) return;
    const u64 k = t / qdeg;
    const u32 p = (u32)(t % qdeg);
    u64* dst = qq2 + t * qdim;
    if (k >= n) { for (u32 c = 0; c < qdim; c++) dst[c] = 0; return; }
    const u64 s = gl::pow(shift_in, p);
    const u64* src = qq1 + ((u64)p * n + k) * qdim;
    for (u32 c = 0; c < qdim; c++) dst[c] = gl::mulc(src[c], s);
}

int dev_q_split(Ctx* ctx, u64* qq2, const u64* qq1, u64 n, u64 n_ext, u32 qdeg, u32 qdim, u64 shift_in)
{
    if (!n_ext || !qdeg || !qdim) return 0;
    if (n > n_ext || (u64)qdeg * n > n_ext) return xlp_fail("q split: needs qDeg * N <= NExtended");
    if (qq2 == qq1) return xlp_fail("q split: in-place is not supported");
    const u64 thr = n_ext * qdeg;
    hipLaunchKernelGGL(q_split_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, ctx->stream, qq2, qq1, n, n_ext, qdeg, qdim, shift_in % hgl::P);
    XLP_HIP(hipGetLastError());
    return 0;
}

// Polinomial::copy between strided views (polinomial.hpp copy(); the transposeH1H2Columns / transposeZColumns /
// ...Rows helpers of starks.cpp:406-555 are sequences of it): dst[i * dst_stride + c] = src[i * src_stride + c], c < dim.
__global__ __launch_bounds__(256) void copy_cols_kernel(u64* dst, u64 dst_stride, const u64* src, u64 src_stride, u64 n, u32 dim)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * dim) return;
    const u64 i = t / dim;
    const u32 c = (u32)(t % dim);
    dst[i * dst_stride + c] = src[i * src_stride + c];
}
int dev_copy_cols(Ctx* ctx, u64* dst, u64 dst_stride, const u64* src, u64 src_stride, u64 n, u32 dim)
{
    if (!n || !dim) return 0;
    if (dst_stride < dim || src_stride < dim) return xlp_fail("copy cols: stride smaller than dim");
    const u64 thr = n * dim;
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, ctx->stream, dst, dst_stride, src, src_stride, n, dim);
    XLP_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ sums of products without carries (evmap, dot_columns)
// One factor is cut into limbs of 22 + 21 + 21 bits, the other into its 32-bit halves; the six partial products (< 2^54) each have an
// accumulator of their own: one v_mad_u64_u32 apiece and no carry chain.  An accumulator takes 1 024 such products (1024 * (2^32 - 1) * (2^22 - 1)
// < 2^64); flush() folds the six into the canonical running sum r and clears them.
struct DcSum {
    u64 a[6];                                         // partial products of weight 2^0, 2^22, 2^43, 2^32, 2^54, 2^75
    u64 r;                                            // what has been folded so far (canonical)
    __device__ __forceinline__ void clear()
    {
        r = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) a[i] = 0;
    }
    // a += (l0 + l1 2^22 + l2 2^43) * v, v any 64-bit word
    __device__ __forceinline__ void mac(u32 l0, u32 l1, u32 l2, u64 v)
    {
        const u32 v0 = (u32)v, v1 = (u32)(v >> 32);
        a[0] += (u64)v0 * l0; a[1] += (u64)v0 * l1; a[2] += (u64)v0 * l2;
        a[3] += (u64)v1 * l0; a[4] += (u64)v1 * l1; a[5] += (u64)v1 * l2;
    }
    __device__ __forceinline__ void flush()
    {
        constexpr u64 W[6] = {1ULL, 1ULL << 22, 1ULL << 43, 1ULL << 32, 1ULL << 54, (1ULL << 43) - (1ULL << 11)};   // 2^75 = 2^11 (2^32 - 1)
#pragma unroll
        for (int i = 0; i < 6; i++) { r = gl::addc(r, gl::mulc(a[i], W[i])); a[i] = 0; }
    }
};

// ------------------------------------------------------------------ evmap
// evals[i] = sum_k L_i[k] * pol_i[(k << eb) * stride_i].  Lanes run along the ENTRIES, sorted by address on the host:
// neighbouring entries are neighbouring columns of one row-major section, so a wave reads contiguous bytes of a
// row (a column-per-block layout gathers 8 bytes per 128-byte line: 15x slower at zkEVM widths).  L[k] is
// wave-uniform per prime flag (one broadcast line).  Blocks cover EV_ROWS rows x EV_T entries and write partial sums
// partial[chunk][sub-entry][3]; evmap_final_kernel adds the chunks.
// An extension column (x0, x1, x2) is entered as THREE base-field sub-entries: L * (x0 + x1 a + x2 a^2) =
// x0 * L + x1 * (a L) + x2 * (a^2 L) with a L = (l2, l0 + l2, l1), a^2 L = (l1, l1 + l2, l0 + l2) -- so every lane runs the
// same unreduced multiply-accumulate (a wave with one extension lane used to drag all 64 through a full Fp^3 product per row).
constexpr int EV_T = 256, EV_ROWS = 4096, EV_SUB = 256, EV_FOLD = 1024;   // rows per block / rows of L staged in LDS at a time / products per accumulator
struct EvEntryDev { const u64* pol; u64 stride; u32 variant; u32 prime; u32 sub; u32 pad; };
typedef const __attribute__((address_space(1))) u64* gptr_t;   // entry pointers come from memory: tell the compiler they are global

// Round 3: the products without carries of DcSum.  The multiplier words of a row (l0, l1, l2, l0 + l2, l1 + l2 of LEv / LpEv) are staged in LDS
// already cut into their three limbs, a lane reads the nine limbs of its three multipliers and its column word and issues 18 multiply-adds --
// the 128-bit products with a compare per addition took ~45 vector instructions per row and lane, as long as the rows take to arrive.
__global__ __launch_bounds__(EV_T) void evmap_partial_kernel(u64* partial, const u64* LEv, const u64* LpEv, u64 n, u32 eb, const EvEntryDev* ents, u32 nsub)
{
    __shared__ u32 sL[2][EV_SUB * 15];                     // per row of LEv / LpEv: five multiplier words x three limbs (30 KiB)
    const u32 ei = blockIdx.x * EV_T + threadIdx.x;          // entry blocks fastest: the blocks of one row range run together
    const bool live = ei < nsub;
    EvEntryDev e = ents[live ? ei : 0];
    const u64 k0 = (u64)blockIdx.y * EV_ROWS;
    const u64 k1 = k0 + EV_ROWS < n ? k0 + EV_ROWS : n;
    const u64 rstep = e.stride << eb;
    gptr_t b = (gptr_t)e.pol + k0 * rstep;
    // the multiplier's components for this lane: L -> (0, 1, 2), a L -> (2, 3, 1), a^2 L -> (1, 4, 3); as dword offsets into a row's 15
    const u32 i0 = 3 * (e.variant == 0 ? 0 : (e.variant == 1 ? 2 : 1)), i1 = 3 * (e.variant == 0 ? 1 : (e.variant == 1 ? 3 : 4)), i2 = 3 * (e.variant == 0 ? 2 : (e.variant == 1 ? 1 : 3));
    const u32* l = sL[e.prime ? 1 : 0];
    DcSum S[3];
    S[0].clear(); S[1].clear(); S[2].clear();
    u32 since = 0;                                          // rows accumulated since the last fold (block-uniform)
    constexpr int U = 8;                                   // rows in flight per lane
    for (u64 s0 = k0; s0 < k1; s0 += EV_SUB) {
        const u32 rows = (u32)(k1 - s0 < EV_SUB ? k1 - s0 : EV_SUB);
        __syncthreads();                                   // previous sub-chunk fully consumed
        for (u32 i = threadIdx.x; i < rows * 2; i += EV_T) {
            const u32 r = i >> 1;
            const u64* src = (i & 1) ? LpEv : LEv;
            u32* d = &sL[i & 1][15 * r];
            const u64 a0 = gl::canon(src[3 * (s0 + r)]), a1 = gl::canon(src[3 * (s0 + r) + 1]), a2 = gl::canon(src[3 * (s0 + r) + 2]);
            const u64 m[5] = {a0, a1, a2, gl::addc(a0, a2), gl::addc(a1, a2)};
#pragma unroll
            for (int q = 0; q < 5; q++) { d[3 * q] = (u32)m[q] & 0x3FFFFFu; d[3 * q + 1] = (u32)(m[q] >> 22) & 0x1FFFFFu; d[3 * q + 2] = (u32)(m[q] >> 43); }
        }
        __syncthreads();
        if (since + rows > EV_FOLD) { S[0].flush(); S[1].flush(); S[2].flush(); since = 0; }
        since += rows;
        if (!live) continue;
        auto row = [&](u32 k, u64 v) {
            const u32* lr = l + 15 * k;
            S[0].mac(lr[i0], lr[i0 + 1], lr[i0 + 2], v);
            S[1].mac(lr[i1], lr[i1 + 1], lr[i1 + 2], v);
            S[2].mac(lr[i2], lr[i2 + 1], lr[i2 + 2], v);
        };
        u32 k = 0;
        for (; k + U <= rows; k += U, b += U * rstep) {
            u64 v[U];
#pragma unroll
            for (int j = 0; j < U; j++) v[j] = b[j * rstep];
#pragma unroll
            for (int j = 0; j < U; j++) row(k + j, v[j]);
        }
        for (; k < rows; k++, b += rstep) row(k, b[0]);
    }
    if (!live) return;
    u64* dst = partial + ((u64)blockIdx.y * nsub + e.sub) * 3;
    for (int c = 0; c < 3; c++) { S[c].flush(); dst[c] = S[c].r; }
}

// evals[i] = sum over the chunks and over the sub-entries first[i] .. first[i + 1] - 1 of entry i
__global__ void evmap_final_kernel(u64* evals, const u64* partial, u64 nchunks, u32 nev, u32 nsub, const u32* first)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nev * 3) return;
    const u32 i = t / 3, c = t % 3;
    u64 s = 0;
    for (u32 q = first[i]; q < first[i + 1]; q++)
        for (u64 ch = 0; ch < nchunks; ch++) s = gl::addc(s, partial[(ch * nsub + q) * 3 + c]);
    evals[t] = s;
}

int dev_evmap(Ctx* ctx, u64* evals, const u64* LEv, const u64* LpEv, u64 n, u32 eb, u32 nev, const xlp_evmap_entry* entries)
{
    if (!nev) return 0;
    if (!n) { XLP_HIP(hipMemsetAsync(evals, 0, (size_t)nev * 24, ctx->stream)); return 0; }
    const u64 nchunks = (n + EV_ROWS - 1) / EV_ROWS;
    std::vector<EvEntryDev> h;
    std::vector<u32> first(nev + 1);
    for (u32 i = 0; i < nev; i++) {
        if (entries[i].dim != 1 && entries[i].dim != 3) return xlp_fail("evmap: dim must be 1 or 3");
        first[i] = (u32)h.size();
        for (u32 v = 0; v < entries[i].dim; v++)
            h.push_back(EvEntryDev{entries[i].pol + v, entries[i].row_stride, v, entries[i].prime, (u32)h.size(), 0});
    }
    const u32 nsub = (u32)h.size();
    first[nev] = nsub;
    std::stable_sort(h.begin(), h.end(), [](const EvEntryDev& x, const EvEntryDev& y) { return x.pol < y.pol; });   // lanes = neighbouring columns
    u64* scratch = nullptr;
    const size_t ent_bytes = ((size_t)nsub * sizeof(EvEntryDev) + 255) / 256 * 256;
    const size_t first_bytes = ((size_t)(nev + 1) * 4 + 255) / 256 * 256;
    if (ctx_scratch(ctx, ent_bytes + first_bytes + nchunks * nsub * 24, 2, &scratch)) return -1;
    XLP_HIP(hipMemcpyAsync(scratch, h.data(), nsub * sizeof(EvEntryDev), hipMemcpyHostToDevice, ctx->stream));
    u32* dfirst = (u32*)((char*)scratch + ent_bytes);
    XLP_HIP(hipMemcpyAsync(dfirst, first.data(), (nev + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    XLP_HIP(hipStreamSynchronize(ctx->stream));   // h, first go out of scope
    u64* partial = (u64*)((char*)scratch + ent_bytes + first_bytes);
    hipLaunchKernelGGL(evmap_partial_kernel, dim3((nsub + EV_T - 1) / EV_T, (unsigned)nchunks), dim3(EV_T), 0, ctx->stream, partial, LEv, LpEv, n, eb,
                       (const EvEntryDev*)scratch, nsub);
    hipLaunchKernelGGL(evmap_final_kernel, dim3((nev * 3 + 63) / 64), dim3(64), 0, ctx->stream, evals, partial, nchunks, nev, nsub, dfirst);
    XLP_HIP(hipGetLastError());
    return 0;
}


// ------------------------------------------------------------------ evmap on column-major sections
// The same sums with the columns stored contiguously (colmajor.hip): coset 0 of an extended column IS the column at the rows (k << eb), so
// an entry is the pointer to its first word (a `prime` entry is multiplied by LpEv instead of LEv, nothing else).  Lanes still run along the ENTRIES in
// the arithmetic (the multiplier limbs of a row are wave-uniform), but the global reads run along the ROWS: a workgroup takes 64 sub-entries,
// each wave loads 64 rows of 16 of them (512 contiguous bytes per instruction), the tile is turned through LDS, and wave w accumulates rows
// 16 w .. 16 w + 15 of the tile for all 64 entries; the four partial sums meet in LDS at the end of the row chunk.
constexpr int EC_T = 256, EC_E = 64, EC_ROWS = 4096, EC_TILE = 64;
struct EvEntryCm { const u64* col; u32 variant; u32 prime; u32 sub; u32 pad; };
__global__ __launch_bounds__(EC_T) void evmap_cm_partial_kernel(u64* partial, const u64* LEv, const u64* LpEv, u64 n, const EvEntryCm* ents, u32 nsub)
{
    __shared__ u32 sL[2][EC_TILE * 15];                    // the multiplier limbs of the tile's rows: [LEv | LpEv][row][5 words x 3 limbs]
    __shared__ u64 tile[EC_TILE][EC_E + 1];
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 e0 = blockIdx.x * EC_E;
    const u32 ei = e0 + lane;
    const bool live = ei < nsub;
    const EvEntryCm e = ents[live ? ei : 0];
    const u64 k0 = (u64)blockIdx.y * EC_ROWS;
    const u64 k1 = k0 + EC_ROWS < n ? k0 + EC_ROWS : n;
    const u32 i0 = 3 * (e.variant == 0 ? 0 : (e.variant == 1 ? 2 : 1)), i1 = 3 * (e.variant == 0 ? 1 : (e.variant == 1 ? 3 : 4)), i2 = 3 * (e.variant == 0 ? 2 : (e.variant == 1 ? 1 : 3));
    const u32* l = sL[e.prime ? 1 : 0];
    // this wave's 16 entries of the tile (lane = row) and, for threads 0 .. 127, one row of LEv / LpEv: fetched one tile ahead
    const u64* mycol[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { const u32 ej = e0 + wv * 16 + j; mycol[j] = ents[ej < nsub ? ej : 0].col; }
    const u64* Lsrc = (threadIdx.x & 1) ? LpEv : LEv;
    u64 v[16], a[3];
    auto fetch = [&](u64 t0) {
        const bool in = t0 + lane < k1;
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = (in && e0 + wv * 16 + j < nsub) ? mycol[j][t0 + lane] : 0;
        const u64 r = t0 + (threadIdx.x >> 1);
        if (threadIdx.x < 2 * EC_TILE && r < k1) { a[0] = Lsrc[3 * r]; a[1] = Lsrc[3 * r + 1]; a[2] = Lsrc[3 * r + 2]; }
        else a[0] = a[1] = a[2] = 0;
    };
    DcSum S[3];
    S[0].clear(); S[1].clear(); S[2].clear();
    u32 since = 0;
    fetch(k0);
    for (u64 t0 = k0; t0 < k1; t0 += EC_TILE) {
        const u32 trows = (u32)(k1 - t0 < EC_TILE ? k1 - t0 : EC_TILE);
        __syncthreads();                                   // the previous tile has been consumed
#pragma unroll
        for (int j = 0; j < 16; j++) tile[lane][wv * 16 + j] = v[j];
        if (threadIdx.x < 2 * EC_TILE) {
            u32* d = &sL[threadIdx.x & 1][15 * (threadIdx.x >> 1)];
            const u64 a0 = gl::canon(a[0]), a1 = gl::canon(a[1]), a2 = gl::canon(a[2]);
            const u64 m[5] = {a0, a1, a2, gl::addc(a0, a2), gl::addc(a1, a2)};
#pragma unroll
            for (int q = 0; q < 5; q++) { d[3 * q] = (u32)m[q] & 0x3FFFFFu; d[3 * q + 1] = (u32)(m[q] >> 22) & 0x1FFFFFu; d[3 * q + 2] = (u32)(m[q] >> 43); }
        }
        __syncthreads();
        if (t0 + EC_TILE < k1) fetch(t0 + EC_TILE);          // in flight while this tile is multiplied
        if (since + 16 > EV_FOLD) { S[0].flush(); S[1].flush(); S[2].flush(); since = 0; }
        since += 16;
        if (live) {
            const u32 r1 = wv * 16 + 16 < trows ? wv * 16 + 16 : trows;
#pragma unroll 4
            for (u32 rr = wv * 16; rr < r1; rr++) {
                const u32* lr = l + 15 * rr;
                const u64 x = tile[rr][lane];
                S[0].mac(lr[i0], lr[i0 + 1], lr[i0 + 2], x);
                S[1].mac(lr[i1], lr[i1 + 1], lr[i1 + 2], x);
                S[2].mac(lr[i2], lr[i2 + 1], lr[i2 + 2], x);
            }
        }
    }
    // the four waves' sums of the same 64 entries
    __syncthreads();
    for (int c = 0; c < 3; c++) { S[c].flush(); tile[wv * 3 + c][lane] = S[c].r; }
    __syncthreads();
    if (wv == 0 && live) {
        u64* dst = partial + ((u64)blockIdx.y * nsub + e.sub) * 3;
        for (int c = 0; c < 3; c++) dst[c] = gl::addc(gl::addc(tile[c][lane], tile[3 + c][lane]), gl::addc(tile[6 + c][lane], tile[9 + c][lane]));
    }
}

// entries: pol = the column's first word (coset 0 of an extended column), row_stride = the words between the components of an extension column
int dev_evmap_cm(Ctx* ctx, u64* evals, const u64* LEv, const u64* LpEv, u64 n, u32 nev, const xlp_evmap_entry* entries)
{
    if (!nev) return 0;
    if (!n) { XLP_HIP(hipMemsetAsync(evals, 0, (size_t)nev * 24, ctx->stream)); return 0; }
    const u64 nchunks = (n + EC_ROWS - 1) / EC_ROWS;
    std::vector<EvEntryCm> h;
    std::vector<u32> first(nev + 1);
    for (u32 i = 0; i < nev; i++) {
        if (entries[i].dim != 1 && entries[i].dim != 3) return xlp_fail("evmap: dim must be 1 or 3");
        first[i] = (u32)h.size();
        for (u32 v = 0; v < entries[i].dim; v++) h.push_back(EvEntryCm{entries[i].pol + v * entries[i].row_stride, v, entries[i].prime ? 1u : 0u, (u32)h.size(), 0});
    }
    const u32 nsub = (u32)h.size();
    first[nev] = nsub;
    std::stable_sort(h.begin(), h.end(), [](const EvEntryCm& x, const EvEntryCm& y) { return x.col < y.col; });   // the two entries of a column (LEv and LpEv) in one tile
    u64* scratch = nullptr;
    const size_t ent_bytes = ((size_t)nsub * sizeof(EvEntryCm) + 255) / 256 * 256;
    const size_t first_bytes = ((size_t)(nev + 1) * 4 + 255) / 256 * 256;
    if (ctx_scratch(ctx, ent_bytes + first_bytes + nchunks * nsub * 24, 2, &scratch)) return -1;
    XLP_HIP(hipMemcpyAsync(scratch, h.data(), nsub * sizeof(EvEntryCm), hipMemcpyHostToDevice, ctx->stream));
    u32* dfirst = (u32*)((char*)scratch + ent_bytes);
    XLP_HIP(hipMemcpyAsync(dfirst, first.data(), (nev + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    XLP_HIP(hipStreamSynchronize(ctx->stream));   // h, first go out of scope
    u64* partial = (u64*)((char*)scratch + ent_bytes + first_bytes);
    hipLaunchKernelGGL(evmap_cm_partial_kernel, dim3((nsub + EC_E - 1) / EC_E, (unsigned)nchunks), dim3(EC_T), 0, ctx->stream, partial, LEv, LpEv, n, (const EvEntryCm*)scratch, nsub);
    hipLaunchKernelGGL(evmap_final_kernel, dim3((nev * 3 + 63) / 64), dim3(64), 0, ctx->stream, evals, partial, nchunks, nev, nsub, dfirst);
    XLP_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ dot_columns on column-major sections
// out column 3 g + c (out_cstride words apart) = component c of sum_e coefs[coef_e] * pol_e over the entries of group g, for the storage rows
// [0, nrows) -- every named column holds at least nrows words.  Lanes are rows here without any transposition: a wave reads 1 KiB of one column
// per instruction (two rows per lane), the coefficient limbs are wave-uniform scalars.  One workgroup = 128 rows, three waves: WAVE = COMPONENT
// of the Fp^3 sums, as in the row-major kernel.  The host groups the entries by COLUMN: a column word is loaded once and multiplied into the
// accumulators of all ngroups groups (limbs zero where a group does not name the column; several entries of one group on one column are summed
// when the records are expanded), so which accumulator a product goes to is static.
constexpr int DCM_T = 256, DCM_ROWS = 256;
struct DcmContrib { u32 col, group, variant, pad; u64 coef; };     // entry -> column record `col`, group, multiplier variant (1, a, a^2), coefficient index
// rec[col]: {ptr lo, ptr hi, pad, pad} then limbs[comp][g][3] (u32), g < 4: 4 + 36 dwords
constexpr u32 DCM_REC = 40;
__global__ void dcm_expand_kernel(u32* rec, const DcmContrib* con, const u32* cfirst, u32 ncol, const u64* coefs)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncol * 12) return;
    const u32 col = t / 12, comp = (t % 12) / 4, g = t % 4;
    u64 k = 0;
    for (u32 i = cfirst[col]; i < cfirst[col + 1]; i++) {
        if (con[i].group != g) continue;
        const u64* c3 = coefs + 3 * con[i].coef;
        const u64 k0 = gl::canon(c3[0]), k1 = gl::canon(c3[1]), k2 = gl::canon(c3[2]);
        // the multiplier of this column word: k, a k = (k2, k0 + k2, k1), a^2 k = (k1, k1 + k2, k0 + k2)
        const u64 m0 = con[i].variant == 0 ? k0 : con[i].variant == 1 ? k2 : k1;
        const u64 m1 = con[i].variant == 0 ? k1 : con[i].variant == 1 ? gl::addc(k0, k2) : gl::addc(k1, k2);
        const u64 m2 = con[i].variant == 0 ? k2 : con[i].variant == 1 ? k1 : gl::addc(k0, k2);
        k = gl::addc(k, comp == 0 ? m0 : comp == 1 ? m1 : m2);
    }
    u32* d = rec + (u64)col * DCM_REC + 4 + (comp * 4 + g) * 3;
    d[0] = (u32)k & 0x3FFFFFu; d[1] = (u32)(k >> 22) & 0x1FFFFFu; d[2] = (u32)(k >> 43);
}

// One lane = one row, ALL three components and all G groups of its sums (18 G accumulators): a column word is loaded once per row and
// meets 3 G wave-uniform multipliers.  The column records (pointer + limbs: 160 bytes each) go through LDS in chunks of DCM_CH columns,
// double-buffered: read with scalar loads they cost a scalar-cache miss per column that three waves per SIMD do not cover (the row-major
// kernel's lesson, profiles/r3/dot_columns_ab.txt); from LDS they are broadcast reads.  The words of the next DCM_U columns are in flight
// while the current ones are multiplied.
constexpr int DCM_U = 4, DCM_CH = 32;
template <int G>
__global__ __launch_bounds__(DCM_T, G <= 3 ? 3 : 2) void dot_columns_cm_kernel(u64* out, u64 out_cstride, const u32* __restrict__ rec, u32 ncol, u64 nrows)
{
    __shared__ u32 srec[2][DCM_CH * DCM_REC];
    static_assert(DCM_CH * DCM_REC == 5 * DCM_T, "a thread moves five dwords of a chunk");
    const u64 row = (u64)blockIdx.x * DCM_ROWS + threadIdx.x;
    const bool ok = row < nrows;
    const u64 rr = ok ? row : nrows - 1;                      // lanes past the end read the last row (in bounds) and store nothing
    DcSum S[3][G];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int g = 0; g < G; g++) S[c][g].clear();
    const u32 nch = (ncol + DCM_CH - 1) / DCM_CH;
    const u64 nwords = (u64)ncol * DCM_REC;
    u32 mv[5];
    auto gfetch = [&](u32 ch) {
#pragma unroll
        for (int i = 0; i < 5; i++) { const u64 w = (u64)ch * DCM_CH * DCM_REC + threadIdx.x + i * DCM_T; mv[i] = w < nwords ? rec[w] : 0; }
    };
    auto lput = [&](u32 buf) {
#pragma unroll
        for (int i = 0; i < 5; i++) srec[buf][threadIdx.x + i * DCM_T] = mv[i];
    };
    gfetch(0); lput(0);
    __syncthreads();
    u32 since = 0;
    for (u32 ch = 0; ch < nch; ch++) {
        if (ch + 1 < nch) gfetch(ch + 1);
        const u32* R = srec[ch & 1];
        const u32 ncc = ncol - ch * DCM_CH < (u32)DCM_CH ? ncol - ch * DCM_CH : (u32)DCM_CH;
        auto colptr = [&](u32 j) -> const u64* {
            const u32* r = R + (j < ncc ? j : ncc - 1) * DCM_REC;
            return (const u64*)(((u64)r[1] << 32) | r[0]);
        };
        u64 cur[DCM_U], nxt[DCM_U];
#pragma unroll
        for (int j = 0; j < DCM_U; j++) cur[j] = colptr(j)[rr];
        for (u32 j0 = 0; j0 < ncc; j0 += DCM_U) {
#pragma unroll
            for (int j = 0; j < DCM_U; j++) nxt[j] = colptr(j0 + DCM_U + j)[rr];
            if (since + DCM_U > 1024) {
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int g = 0; g < G; g++) S[c][g].flush();
                since = 0;
            }
            since += DCM_U;
#pragma unroll
            for (int j = 0; j < DCM_U; j++) {
                if (j0 + j >= ncc) break;
                const u32* r = R + (j0 + j) * DCM_REC + 4;
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int g = 0; g < G; g++) S[c][g].mac(r[(c * 4 + g) * 3], r[(c * 4 + g) * 3 + 1], r[(c * 4 + g) * 3 + 2], cur[j]);
            }
#pragma unroll
            for (int j = 0; j < DCM_U; j++) cur[j] = nxt[j];
        }
        if (ch + 1 < nch) lput((ch + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int g = 0; g < G; g++) {
            S[c][g].flush();
            if (ok) out[(u64)(3 * g + c) * out_cstride + row] = S[c][g].r;
        }
}

// entries: pol = the column's first word, row_stride = the words between the components of an extension column (dim 3)
int dev_dot_columns_cm(Ctx* ctx, u64* out, u64 out_cstride, const xlp_dot_entry* entries, u64 nentries, u32 ngroups, const u64* coefs, u64 coefs_size, u64 nrows)
{
    if (ngroups < 1 || ngroups > 4) return xlp_fail("dot columns: 1..4 groups");
    if (!nrows) return 0;
    if (out_cstride < nrows) return xlp_fail("dot columns (column-major): output column stride smaller than the rows");
    std::vector<std::pair<const u64*, DcmContrib>> all;
    for (u64 i = 0; i < nentries; i++) {
        const xlp_dot_entry& x = entries[i];
        if (x.dim != 1 && x.dim != 3) return xlp_fail("dot columns: dim must be 1 or 3");
        if (x.group >= ngroups) return xlp_fail("dot columns: group out of range");
        if (x.coef * 3 + 3 > coefs_size) return xlp_fail("dot columns: coefficient index out of range");
        for (u32 v = 0; v < x.dim; v++) all.push_back({x.pol + v * x.row_stride, DcmContrib{0, x.group, v, 0, x.coef}});
    }
    std::stable_sort(all.begin(), all.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    std::vector<DcmContrib> con;
    std::vector<u32> cfirst;
    std::vector<u32> rec;
    for (size_t i = 0; i < all.size(); i++) {
        if (i == 0 || all[i].first != all[i - 1].first) {
            cfirst.push_back((u32)con.size());
            const u64 ptr = (u64)(uintptr_t)all[i].first;
            rec.resize(rec.size() + DCM_REC, 0);
            rec[rec.size() - DCM_REC] = (u32)ptr; rec[rec.size() - DCM_REC + 1] = (u32)(ptr >> 32);
        }
        DcmContrib c = all[i].second;
        c.col = (u32)cfirst.size() - 1;
        con.push_back(c);
    }
    const u32 ncol = (u32)cfirst.size();
    cfirst.push_back((u32)con.size());
    if (!ncol) {                                          // no entries: every sum is zero
        for (u32 k = 0; k < 3 * ngroups; k++) XLP_HIP(hipMemsetAsync(out + (u64)k * out_cstride, 0, nrows * sizeof(u64), ctx->stream));
        return 0;
    }
    u64* scratch = nullptr;
    const size_t rec_b = ((size_t)rec.size() * 4 + 255) / 256 * 256, con_b = ((size_t)con.size() * sizeof(DcmContrib) + 255) / 256 * 256, cf_b = ((size_t)cfirst.size() * 4 + 255) / 256 * 256;
    if (ctx_scratch(ctx, rec_b + con_b + cf_b, 2, &scratch)) return -1;
    u32* drec = (u32*)scratch;
    DcmContrib* dcon = (DcmContrib*)((char*)scratch + rec_b);
    u32* dcf = (u32*)((char*)scratch + rec_b + con_b);
    XLP_HIP(hipMemcpyAsync(drec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    XLP_HIP(hipMemcpyAsync(dcon, con.data(), con.size() * sizeof(DcmContrib), hipMemcpyHostToDevice, ctx->stream));
    XLP_HIP(hipMemcpyAsync(dcf, cfirst.data(), cfirst.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    XLP_HIP(hipStreamSynchronize(ctx->stream));            // the host vectors go out of scope
    hipLaunchKernelGGL(dcm_expand_kernel, dim3((ncol * 12 + 255) / 256), dim3(256), 0, ctx->stream, drec, dcon, dcf, ncol, coefs);
    const dim3 grid((unsigned)((nrows + DCM_ROWS - 1) / DCM_ROWS)), block(DCM_T);
    switch (ngroups) {
    case 1: hipLaunchKernelGGL(dot_columns_cm_kernel<1>, grid, block, 0, ctx->stream, out, out_cstride, drec, ncol, nrows); break;
    case 2: hipLaunchKernelGGL(dot_columns_cm_kernel<2>, grid, block, 0, ctx->stream, out, out_cstride, drec, ncol, nrows); break;
    case 3: hipLaunchKernelGGL(dot_columns_cm_kernel<3>, grid, block, 0, ctx->stream, out, out_cstride, drec, ncol, nrows); break;
    default: hipLaunchKernelGGL(dot_columns_cm_kernel<4>, grid, block, 0, ctx->stream, out, out_cstride, drec, ncol, nrows); break;
    }
    XLP_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ dot_columns
// out[row][g] = sum_e coefs[coef_e] * pol_e[row]  (Fp^3 coefficient x column of dimension 1 or 3), e over the entries of group g.
// The linear part of the FRI polynomial (starks.cpp:371-380) once its Horner chains are dot products (expr_opt.py): at the zkEVM
// geometry 3 788 (coefficient, column) pairs per row over 1 556 columns, 2^24 rows -- 188 GB read once, and 3 x 3 788 base-field
// products per row, which is what bounds it (VALU).
// One workgroup = DC_ROWS rows, three waves: lane = row, WAVE = COMPONENT of the Fp^3 sums (each wave multiplies the same column
// words by its own word of the coefficients: a lane keeps 6 accumulators per group instead of 18).  The host cuts every section into
// windows of <= DC_CB consecutive words that cover the requested columns; per window the workgroup loads a DC_ROWS x DC_CB tile (one
// wave instruction = 512 contiguous bytes of one row), stores it transposed in LDS, and every wave walks ITS stream of the window's
// items: one 16-byte record per item.
// PRODUCTS WITHOUT CARRIES (round 3): a coefficient word is cut into limbs of 22 + 21 + 21 bits (once, by dot_expand_kernel), the
// column value into its two 32-bit halves; the six partial products (< 2^54) each have an accumulator of their own, one v_mad_u64_u32
// apiece and nothing else -- 6 vector instructions per base-field product where the 128-bit product with its carry chain into a 192-bit
// sum took 14.  An accumulator holds 1 024 such products before it can overflow: the host orders a window's items by group and counts
// them, and a group's accumulators are folded into its field-element sum (DcSum::flush) before the count would pass that.  The items of
// a group run in a loop of their own, so which accumulators a product goes to is static (the first version selected them per item: a
// third of its instructions were register moves and selects).
// THE RECORDS GO THROUGH LDS TOO: read with scalar loads (they are wave-uniform) every batch of four cost a scalar-cache miss of ~450
// cycles that three waves per SIMD do not cover, prefetched one batch ahead or not (SQ_WAIT_ANY 45 % of the wave cycles, VALU 38 % busy:
// profiles/r3/dot_columns_ab.txt).  The workgroup now copies a window's three streams into LDS beside the tile (prefetched with it) and a
// wave reads its records back as broadcast ds_read_b128 -- in-order returns, so the loop (DcPipe) keeps records two batches and column
// words one batch ahead of the multiply-adds with counted waits; the limbs are then vector operands, which v_mad_u64_u32 takes as well.
constexpr int DC_ROWS = 64, DC_CB = 64, DC_T = 192, DC_IB = 4;   // rows per workgroup, words per window, threads, records per batch
constexpr int DC_PRE = (DC_ROWS + 2) / 3;             // rows of a tile each wave loads (wave w: rows w, w + 3, ...)
constexpr u32 DC_HDR = 4;                             // records of a window header
constexpr u32 DC_SMAX = 144;                          // records of one stream of a window (host); three streams + the tile = 40 192 bytes of LDS: four workgroups per CU
constexpr int DC_PREREC = (3 * DC_SMAX + DC_T - 1) / DC_T;
constexpr u32 DC_UNITS = 1024;                        // products one accumulator takes: 1024 * (2^32 - 1) * (2^22 - 1) < 2^64
constexpr u32 DC_DIM3_UNITS = 3;                      // an extension column puts 3 products into a component's accumulators
constexpr u32 DC_COLB = (DC_ROWS + 1) * 8;            // bytes between two columns of the LDS tile
typedef u32 dc_u4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) dc_u4* dc_hdr_t;   // window headers: constant address space = scalar loads
typedef const __attribute__((address_space(1))) dc_u4* dc_rec_t;


// a record: {byte offset of the column in the LDS tile, l0, l1, l2 | group << 24}; in LDS the offsets and the limbs are two arrays.
// Every array of the pipeline is double-buffered with period two -- sent for in one step, used in the next -- so two steps with the
// buffer names swapped (step2) need no register moves; a lone step rotates.
typedef u32 dc_u3 __attribute__((ext_vector_type(3)));
struct DcHalf { u32 m[DC_IB]; dc_u3 l[DC_IB]; u64 v[DC_IB]; };   // m: offsets of the batch AFTER the one l and v belong to
struct DcPipe {
    DcHalf A, B;                                      // between steps: A.l, A.v = batch j (A.v maybe still in flight), B.m = offsets of batch j + 1
    u32 at;                                           // record index of batch j + 2 within the LDS arrays (wave-uniform)
    const u32* metas;
    const u32* limbs;
    const char* tl;                                   // this lane's row of the tile
    __device__ __forceinline__ void load_m(DcHalf& h, u32 rec) const
    {
        const dc_u4 t = *(const dc_u4*)(metas + rec);
        h.m[0] = t.x; h.m[1] = t.y; h.m[2] = t.z; h.m[3] = t.w;
    }
    __device__ __forceinline__ void load_l(DcHalf& h, u32 rec) const
    {
        const dc_u4* q = (const dc_u4*)(limbs + 3 * rec);
        const dc_u4 t0 = q[0], t1 = q[1], t2 = q[2];
        h.l[0] = dc_u3{t0.x, t0.y, t0.z}; h.l[1] = dc_u3{t0.w, t1.x, t1.y}; h.l[2] = dc_u3{t1.z, t1.w, t2.x}; h.l[3] = dc_u3{t2.y, t2.z, t2.w};
    }
    __device__ __forceinline__ void load_v(DcHalf& h, const u32 (&m)[DC_IB]) const
    {
#pragma unroll
        for (int j = 0; j < DC_IB; j++) h.v[j] = *(const u64*)(tl + m[j]);
    }
    __device__ __forceinline__ void start(u32 first, const u32* metas_, const u32* limbs_, const char* tile_lane)
    {
        metas = metas_; limbs = limbs_; tl = tile_lane;
        DcHalf t;
        load_m(t, first);
        load_l(A, first);
        load_m(B, first + DC_IB);
        load_v(A, t.m);
        at = first + 2 * DC_IB;
    }
    // multiplies batch j (in `cur`), sends for the words and limbs of batch j + 1 (into `nxt`, whose offsets it holds) and the offsets of batch j + 2
    __device__ __forceinline__ void half(DcSum& s, DcHalf& cur, DcHalf& nxt)
    {
        load_v(nxt, nxt.m);
        load_l(nxt, at - DC_IB);
        load_m(cur, at);
        at += DC_IB;
#pragma unroll
        for (int j = 0; j < DC_IB; j++) s.mac(cur.l[j].x, cur.l[j].y, cur.l[j].z, cur.v[j]);
    }
    __device__ __forceinline__ void step2(DcSum& s) { half(s, A, B); half(s, B, A); }
    __device__ __forceinline__ void step1(DcSum& s)
    {
        half(s, A, B);
#pragma unroll
        for (int j = 0; j < DC_IB; j++) { const u32 m = A.m[j]; A.l[j] = B.l[j]; A.v[j] = B.v[j]; B.m[j] = m; }
    }
};

template <int G>
__global__ __launch_bounds__(DC_T, 3) void dot_columns_kernel(u64* out, const dc_u4* plan, u32 nwindows, u64 nrows)
{
    constexpr u32 NREC = 3 * DC_SMAX + 2 * DC_IB;         // the three streams back to back; the pipeline reads up to two batches past a stream
    __shared__ u64 tile[DC_CB][DC_ROWS + 1];
    __shared__ __attribute__((aligned(16))) u32 metas[NREC];
    __shared__ __attribute__((aligned(16))) u32 limbs[3 * NREC];
    static_assert(sizeof(tile) == DC_CB * DC_COLB, "tile layout");
    const u32 lane = threadIdx.x & 63;
    const u32 comp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: which word of the Fp^3 sums this wave computes
    const u64 row0 = (u64)blockIdx.x * DC_ROWS;
    DcSum S[G];
    u32 units[G];                                          // wave-uniform: products in S[g]'s accumulators since its last flush
#pragma unroll
    for (int g = 0; g < G; g++) { S[g].clear(); units[g] = 0; }
    for (u32 i = threadIdx.x; i < NREC; i += DC_T) metas[i] = 0;   // whatever the pipeline picks up past a stream is then a valid offset: an older record's, or 0
    // The column words and the records of the NEXT window are fetched into registers while the current one is consumed from LDS.
    u64 pre[DC_PRE];
    dc_u4 prerec[DC_PREREC];
    auto fetch = [&](dc_hdr_t hdr) {
        const dc_u4 h0 = hdr[0], h1 = hdr[1];
        const u64 stride = gl::pack(h0.z, h0.w);
        gptr_t src = (gptr_t)gl::pack(h0.x, h0.y) + (row0 + comp) * stride + lane;
        const u32 ncols = h1.x, nrec3 = 3 * h1.y;
#pragma unroll
        for (int i = 0; i < DC_PRE; i++)
            pre[i] = (lane < ncols && comp + 3 * i < (u32)DC_ROWS && row0 + comp + 3 * i < nrows) ? src[(u64)(3 * i) * stride] : 0;
        dc_rec_t rsrc = (dc_rec_t)(const dc_u4*)(hdr + DC_HDR);
#pragma unroll
        for (int i = 0; i < DC_PREREC; i++)
            prerec[i] = threadIdx.x + i * DC_T < nrec3 ? rsrc[threadIdx.x + i * DC_T] : dc_u4{0, 0, 0, 0};
    };
    dc_hdr_t p = (dc_hdr_t)plan;
    if (nwindows) fetch(p);
    for (u32 w = 0; w < nwindows; w++) {
        const dc_u4 h1 = p[1];
        const u32 nrec = h1.y, n3 = h1.w >> 16, wunits = n3 * DC_DIM3_UNITS;   // records per stream; items of dimension 3
        const u64 counts = gl::pack(h1.z, h1.w);           // items of dimension 1 per group, padded to DC_IB (4 x 12 bits)
        p += DC_HDR + 3 * nrec;
        __syncthreads();                                   // previous window fully consumed
#pragma unroll
        for (int i = 0; i < DC_PRE; i++)
            if (comp + 3 * i < (u32)DC_ROWS) tile[lane][comp + 3 * i] = pre[i];
#pragma unroll
        for (int i = 0; i < DC_PREREC; i++) {
            const u32 k = threadIdx.x + i * DC_T;
            if (k < 3 * DC_SMAX) { metas[k] = prerec[i].x; limbs[3 * k] = prerec[i].y; limbs[3 * k + 1] = prerec[i].z; limbs[3 * k + 2] = prerec[i].w; }
        }
        __syncthreads();
        if (w + 1 < nwindows) fetch(p);
        DcPipe pipe;
        pipe.start(comp * nrec, metas, limbs, (const char*)&tile[0][lane]);
        u32 n1all = 0;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const u32 n1 = (u32)(counts >> (12 * g)) & 0xFFF;
            if (units[g] + n1 + wunits > DC_UNITS) { S[g].flush(); units[g] = 0; }
            units[g] += n1 + wunits;
            u32 i = 0;
            for (; i + 2 * DC_IB <= n1; i += 2 * DC_IB) pipe.step2(S[g]);
            if (i < n1) pipe.step1(S[g]);
            n1all += n1;
        }
        // extension columns: (k0, k1, k2) * (x0, x1, x2) in Fp[X] / (X^3 - X - 1), the nine products regrouped as
        // c0 = k0 x0 + k1 x2 + k2 x1, c1 = k0 x1 + k1 (x0 + x2) + k2 (x1 + x2), c2 = k0 x2 + k1 x1 + k2 (x0 + x2) (the two sums of column
        // words taken in the field): three records (k0, k1, k2) and three products per component.  No real FRI polynomial has such an
        // item (expr_opt.py splits extension columns into their words and folds X, X^2 into the coefficient table); the entry point
        // accepts them, and they take this plain loop behind the pipelined one
        const u32 q3 = comp * nrec + n1all;
        for (u32 i = 0; i < n3; i++) {
            const u32 k = q3 + 3 * i;
            const char* tc = (const char*)&tile[0][lane] + metas[k];
            const u32 g = __builtin_amdgcn_readfirstlane(limbs[3 * k + 2] >> 24);
            const u64 x0 = gl::canon(*(const u64*)tc), x1 = gl::canon(*(const u64*)(tc + DC_COLB)), x2 = gl::canon(*(const u64*)(tc + 2 * DC_COLB));
            const u64 x02 = gl::addc(x0, x2), x12 = gl::addc(x1, x2);
            const u64 xa = comp == 0 ? x0 : comp == 1 ? x1 : x2, xb = comp == 0 ? x2 : comp == 1 ? x02 : x1, xc = comp == 0 ? x1 : comp == 1 ? x12 : x02;
#pragma unroll
            for (int gg = 0; gg < G; gg++)
                if (g == (u32)gg) {
                    S[gg].mac(limbs[3 * k], limbs[3 * k + 1], limbs[3 * k + 2] & 0x1FFFFFu, xa);
                    S[gg].mac(limbs[3 * k + 3], limbs[3 * k + 4], limbs[3 * k + 5] & 0x1FFFFFu, xb);
                    S[gg].mac(limbs[3 * k + 6], limbs[3 * k + 7], limbs[3 * k + 8] & 0x1FFFFFu, xc);
                }
        }
    }
    if (row0 + lane >= nrows) return;
    u64* dst = out + (row0 + lane) * (3 * G) + comp;
#pragma unroll
    for (int g = 0; g < G; g++) {
        S[g].flush();
        dst[3 * g] = S[g].r;
    }
}

// item records {meta, coefficient index, word of the coefficient, ~0} become {meta, l0, l1, l2}: the limbs of that word
__global__ void dot_expand_kernel(dc_u4* plan, u64 nrecords, const u64* coefs)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrecords) return;
    const dc_u4 r = plan[i];
    if (r.w != 0xFFFFFFFFu) return;                        // a window header or padding (a third limb has 21 bits, a header's last word 28)
    const u64 k = coefs[3 * (u64)r.y + (r.z & 3)];         // r.z: word of the coefficient | group << 24 (records of dimension-3 items)
    plan[i] = dc_u4{r.x, (u32)k & 0x3FFFFFu, (u32)(k >> 22) & 0x1FFFFFu, (u32)(k >> 43) | (r.z & 0xFF000000u)};
}

int dev_dot_columns(Ctx* ctx, u64* out, const xlp_dot_entry* entries, u64 nentries, u32 ngroups, const u64* coefs, u64 coefs_size, u64 nrows)
{
    if (ngroups > 4) return xlp_fail("dot columns: at most 4 groups");
    if (coefs_size >> 32) return xlp_fail("dot columns: coefficient table too large");
    std::vector<xlp_dot_entry> e(entries, entries + nentries);
    for (const auto& x : e) {
        if (x.dim != 1 && x.dim != 3) return xlp_fail("dot columns: dim must be 1 or 3");
        if (x.group >= ngroups) return xlp_fail("dot columns: group out of range");
        if (x.coef * 3 + 3 > coefs_size) return xlp_fail("dot columns: coefficient index out of range");
        if (x.row_stride < x.dim) return xlp_fail("dot columns: stride smaller than dim");
    }
    std::stable_sort(e.begin(), e.end(), [](const xlp_dot_entry& a, const xlp_dot_entry& b) { return a.pol < b.pol; });
    // windows: runs of entries of one row span (same stride, < DC_CB words from the window's first word, inside one row), closed early
    // when a stream would get longer than DC_SMAX records.  A window = header, then one record stream per component: the
    // dimension-1 items group by group, each run padded to DC_IB records (zero limbs), then three records per dimension-3 item.
    struct Rec { u32 x, y, z, w; };
    std::vector<Rec> plan;
    u32 nwindows = 0;
    for (size_t i = 0; i < e.size();) {
        const u64* w0 = e[i].pol;
        size_t j = i;
        u32 ncols = 0, n3 = 0, n1[4] = {0, 0, 0, 0};
        auto stream_len = [&](const u32* c1, u32 c3) { u32 r = (3 * c3 + DC_IB - 1) / DC_IB * DC_IB; for (u32 g = 0; g < 4; g++) r += (c1[g] + DC_IB - 1) / DC_IB * DC_IB; return r; };
        while (j < e.size() && e[j].row_stride == e[i].row_stride && (u64)(e[j].pol - w0) + e[j].dim <= (u64)DC_CB &&
               (u64)(e[j].pol - w0) + e[j].dim <= e[i].row_stride) {
            u32 t1[4] = {n1[0], n1[1], n1[2], n1[3]}, t3 = n3;
            if (e[j].dim == 3) t3++; else t1[e[j].group]++;
            if (j > i && stream_len(t1, t3) > DC_SMAX) break;
            n3 = t3;
            for (u32 g = 0; g < 4; g++) n1[g] = t1[g];
            ncols = std::max<u32>(ncols, (u32)(e[j].pol - w0) + e[j].dim);
            j++;
        }
        // every stream is a whole number of batches, the dimension-3 tail included: the streams start at comp * nrec and are read as 16-byte vectors
        u32 padded[4], nrec = (3 * n3 + DC_IB - 1) / DC_IB * DC_IB;
        for (u32 g = 0; g < 4; g++) { padded[g] = (n1[g] + DC_IB - 1) / DC_IB * DC_IB; nrec += padded[g]; }
        const u64 base = (u64)(uintptr_t)w0, stride = e[i].row_stride;
        const u64 counts = (u64)padded[0] | ((u64)padded[1] << 12) | ((u64)padded[2] << 24) | ((u64)padded[3] << 36) | ((u64)n3 << 48);
        plan.push_back({(u32)base, (u32)(base >> 32), (u32)stride, (u32)(stride >> 32)});
        plan.push_back({ncols, nrec, (u32)counts, (u32)(counts >> 32)});
        for (u32 k = 2; k < DC_HDR; k++) plan.push_back({0, 0, 0, 0});
        for (u32 comp = 0; comp < 3; comp++) {
            for (u32 g = 0; g < 4; g++) {
                u32 put = 0;
                for (size_t k = i; k < j; k++)
                    if (e[k].dim == 1 && e[k].group == g) { plan.push_back({(u32)(e[k].pol - w0) * DC_COLB, (u32)e[k].coef, comp, 0xFFFFFFFFu}); put++; }
                for (; put < padded[g]; put++) plan.push_back({0, 0, 0, 0});          // 0 * word 0 of the window
            }
            for (size_t k = i; k < j; k++)
                if (e[k].dim == 3)
                    for (u32 word = 0; word < 3; word++) plan.push_back({(u32)(e[k].pol - w0) * DC_COLB, (u32)e[k].coef, word | (e[k].group << 24), 0xFFFFFFFFu});
            for (u32 put = 3 * n3; put % DC_IB; put++) plan.push_back({0, 0, 0, 0});   // (never read: the kernel walks n3 items)
        }
        nwindows++;
        i = j;
    }
    u64* dplan = nullptr;
    if (!nrows) return 0;
    if (ctx_scratch(ctx, std::max<size_t>(plan.size() * sizeof(Rec), 16), 2, &dplan)) return -1;
    if (!plan.empty()) {                                   // no entries: every sum is zero (the kernel below writes them)
        XLP_HIP(hipMemcpyAsync(dplan, plan.data(), plan.size() * sizeof(Rec), hipMemcpyHostToDevice, ctx->stream));
        XLP_HIP(hipStreamSynchronize(ctx->stream));        // plan goes out of scope
        hipLaunchKernelGGL(dot_expand_kernel, dim3((unsigned)((plan.size() + 255) / 256)), dim3(256), 0, ctx->stream, (dc_u4*)dplan, (u64)plan.size(), coefs);
    }
    const dim3 grid((unsigned)((nrows + DC_ROWS - 1) / DC_ROWS)), block(DC_T);
    const dc_u4* dp = (const dc_u4*)dplan;
    switch (ngroups) {
    case 1: hipLaunchKernelGGL(dot_columns_kernel<1>, grid, block, 0, ctx->stream, out, dp, nwindows, nrows); break;
    case 2: hipLaunchKernelGGL(dot_columns_kernel<2>, grid, block, 0, ctx->stream, out, dp, nwindows, nrows); break;
    case 3: hipLaunchKernelGGL(dot_columns_kernel<3>, grid, block, 0, ctx->stream, out, dp, nwindows, nrows); break;
    default: hipLaunchKernelGGL(dot_columns_kernel<4>, grid, block, 0, ctx->stream, out, dp, nwindows, nrows); break;
    }
    XLP_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ calculateH1H2
// Polinomial::calculateH1H2_opt1 / _opt3 (polinomial.hpp:349-585; called starks.cpp:108-125).
// The reference walks hash chains sequentially; the observable result only depends on
//   last[key] = the last index of t holding `key`,  count[i] = 1 + #{j : last[f[j]] == i},
//   merged    = t[i] repeated count[i] times, i ascending (length 2n);  h1[k] = merged[2k], h2[k] = merged[2k+1].
// GPU form: open-addressing table (slot owner = one t index of the key, atomicMax of the index),
// per-f lookup + atomicAdd, exclusive scan of the counts, then ONE THREAD PER OUTPUT ELEMENT that
// binary-searches its source index -- balanced however skewed the counts are.
constexpr u32 H_EMPTY = 0xFFFFFFFFu;

template <int DIM>
__device__ __forceinline__ void load_key(u64 (&k)[3], const u64* p, u64 stride, u64 i)
{
    k[0] = gl::canon(p[i * stride]);
    k[1] = DIM == 3 ? gl::canon(p[i * stride + 1]) : 0;
    k[2] = DIM == 3 ? gl::canon(p[i * stride + 2]) : 0;
}
__device__ __forceinline__ u32 hash_key(const u64 (&k)[3], u32 mask)
{
    u64 h = k[0] * 0x9E3779B97F4A7C15ULL ^ (k[1] + 0x7F4A7C15ULL) * 0xBF58476D1CE4E5B9ULL ^ (k[2] + 0x94D049BBULL) * 0x94D049BB133111EBULL;
    h ^= h >> 29;
    return (u32)h & mask;
}

template <int DIM>
__global__ void h1h2_insert_kernel(u32* owner, u32* last, const u64* t, u64 ts, u64 n, u32 mask)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 k[3], o[3];
    load_key<DIM>(k, t, ts, i);
    u32 slot = hash_key(k, mask);
    for (;;) {
        u32 cur = atomicCAS(&owner[slot], H_EMPTY, (u32)i);
        if (cur == H_EMPTY) cur = (u32)i;
        load_key<DIM>(o, t, ts, cur);
        if (o[0] == k[0] && o[1] == k[1] && o[2] == k[2]) { atomicMax(&last[slot], (u32)i); return; }
        slot = (slot + 1) & mask;
    }
}

template <int DIM>
__global__ void h1h2_count_kernel(u32* count, u32* err, const u32* owner, const u32* last, const u64* f, u64 fs, const u64* t, u64 ts, u64 n, u32 mask)
{
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    u64 k[3], o[3];
    load_key<DIM>(k, f, fs, j);
    u32 slot = hash_key(k, mask);
    for (;;) {
        const u32 cur = owner[slot];
        if (cur == H_EMPTY) { atomicExch(err, 1u); return; }     // number not included (reference: exitProcess)
        load_key<DIM>(o, t, ts, cur);
        if (o[0] == k[0] && o[1] == k[1] && o[2] == k[2]) { atomicAdd(&count[last[slot]], 1u); return; }
        slot = (slot + 1) & mask;
    }
}

__global__ void fill_u32_kernel(u32* p, u64 n, u32 v)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// exclusive scan of u32 counts (n <= 2^26): per-block scan + block sums + apply
constexpr int US_T = 256, US_E = 16, US_B = US_T * US_E;
__global__ __launch_bounds__(US_T) void uscan_block_kernel(u32* out, u32* bsum, const u32* in, u64 n)
{
    __shared__ u32 sh[US_T];
    const u64 base = (u64)blockIdx.x * US_B + (u64)threadIdx.x * US_E;
    u32 v[US_E], acc = 0;
#pragma unroll
    for (int k = 0; k < US_E; k++) { v[k] = acc; acc += base + k < n ? in[base + k] : 0; }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 1; off < US_T; off <<= 1) {
        u32 a = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += a;
        __syncthreads();
    }
    const u32 pre = threadIdx.x ? sh[threadIdx.x - 1] : 0;
#pragma unroll
    for (int k = 0; k < US_E; k++) if (base + k < n) out[base + k] = pre + v[k];
    if (threadIdx.x == US_T - 1 && bsum) bsum[blockIdx.x] = sh[US_T - 1];
}
__global__ void uscan_apply_kernel(u32* out, const u32* boff, u64 n)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += boff[i / US_B];
}
static int uscan_exclusive(Ctx* ctx, u32* out, const u32* in, u64 n, u32* scratch)
{
    const u64 nb = (n + US_B - 1) / US_B;
    hipLaunchKernelGGL(uscan_block_kernel, dim3((unsigned)nb), dim3(US_T), 0, ctx->stream, out, nb > 1 ? scratch : nullptr, in, n);
    if (nb > 1) {
        u32* boff = scratch + nb;
        if (uscan_exclusive(ctx, boff, scratch, nb, boff + nb)) return -1;
        hipLaunchKernelGGL(uscan_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, out, boff, n);
    }
    XLP_HIP(hipGetLastError());
    return 0;
}

template <int DIM>
__global__ void h1h2_expand_kernel(u64* h1, u64 h1s, u64* h2, u64 h2s, const u64* t, u64 ts, const u32* offs, u64 n)
{
    const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= 2 * n) return;
    // largest i with offs[i] <= pos   (offs ascending, offs[0] = 0, every count >= 1)
    u64 lo = 0, hi = n;
    while (hi - lo > 1) { const u64 m = (lo + hi) >> 1; if (offs[m] <= pos) lo = m; else hi = m; }
    u64* dst = (pos & 1) ? h2 + (pos >> 1) * h2s : h1 + (pos >> 1) * h1s;
#pragma unroll
    for (int d = 0; d < DIM; d++) dst[d] = gl::canon(t[lo * ts + d]);
}

template <int DIM>
static int h1h2_impl(Ctx* ctx, u64* h1, u64 h1s, u64* h2, u64 h2s, const u64* f, u64 fs, const u64* t, u64 ts_, u64 n)
{
    u32 ts = 1;
    while (ts < 2 * n) ts <<= 1;
    const u32 mask = ts - 1;
    // scratch: owner[ts] last[ts] count[n] offs[n] err[1] scan[..]
    const u64 nb = (n + US_B - 1) / US_B;
    const size_t words = (size_t)2 * ts + 2 * n + 64 + 4 * (nb + 64);
    u64* raw = nullptr;
    if (ctx_scratch(ctx, words * sizeof(u32), 3, &raw)) return -1;
    u32* owner = (u32*)raw; u32* last = owner + ts; u32* count = last + ts; u32* offs = count + n; u32* err = offs + n; u32* scan = err + 64;
    XLP_HIP(hipMemsetAsync(owner, 0xFF, (size_t)ts * 4, ctx->stream));
    XLP_HIP(hipMemsetAsync(last, 0, (size_t)ts * 4, ctx->stream));
    XLP_HIP(hipMemsetAsync(err, 0, 4, ctx->stream));
    const unsigned g = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(g), dim3(256), 0, ctx->stream, count, n, 1u);
    hipLaunchKernelGGL(h1h2_insert_kernel<DIM>, dim3(g), dim3(256), 0, ctx->stream, owner, last, t, ts_, n, mask);
    hipLaunchKernelGGL(h1h2_count_kernel<DIM>, dim3(g), dim3(256), 0, ctx->stream, count, err, owner, last, f, fs, t, ts_, n, mask);
    XLP_HIP(hipGetLastError());
    u32 herr = 0;
    XLP_HIP(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream));
    XLP_HIP(hipStreamSynchronize(ctx->stream));
    if (herr) return xlp_fail("calculateH1H2: number not included");
    if (uscan_exclusive(ctx, offs, count, n, scan)) return -1;
    hipLaunchKernelGGL(h1h2_expand_kernel<DIM>, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, ctx->stream, h1, h1s, h2, h2s, t, ts_, offs, n);
    XLP_HIP(hipGetLastError());
    return 0;
}

// element i of a polynomial at p[i * stride .. + dim): Polinomial views of row-major sections, as for dev_calculate_z
int dev_calculate_h1h2(Ctx* ctx, u64* h1, u64 h1s, u64* h2, u64 h2s, const u64* f, u64 fs, const u64* t, u64 ts, u64 n, u32 dim)
{
    if (!n) return 0;
    if (n > (1ull << 30)) return xlp_fail("calculateH1H2: too many rows");
    if (h1s < dim || h2s < dim || fs < dim || ts < dim) return xlp_fail("calculateH1H2: stride below the dimension");
    if (dim == 1) return h1h2_impl<1>(ctx, h1, h1s, h2, h2s, f, fs, t, ts, n);
    if (dim == 3) return h1h2_impl<3>(ctx, h1, h1s, h2, h2s, f, fs, t, ts, n);
    return xlp_fail("calculateH1H2: dim must be 1 or 3");
}

}  // namespace xlp
