// FRI folding, transposition and synthetic-trace fill kernels for gfx950.
//
// Replaces the per-group loop of FRIProve::prove (src/starkpil/fri/friProve.cpp:44-108,
// polMulAxi :192, evalPol :201) and FRIProve::getTransposed (:252-271).
//
// One lane folds one group g: it gathers the nX = 2^red strided ext-3 values
// pol[i*M+g], runs the size-nX inverse NTT in registers (base-field twiddles,
// component-wise), scales coefficient i by (shift^-1 * w(bits)^-g)^i and
// evaluates at the challenge x by Horner in the cubic extension.  Adjacent
// lanes read adjacent groups, so every gather instruction is a contiguous
// 24-byte-per-lane run.  HBM-bound (24 B per input element) and tiny next to
// the LDE / Merkle kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gl_field.hpp"
#include "xlp_internal.hpp"

namespace xlp {
using gl::e3;
using gl::u32;
using gl::u64;

struct FoldParams {
    u64* pol2;
    const u64* pol;
    u32 bits, red;
    u64 x[3];
    u64 shift_inv;
    u64 winv_bits;     // w(bits)^-1
    u64 winv_red;      // w(red)^-1   (root of the size-nX INTT)
    u64 nx_inv;        // 1 / nX
};

template <int RED>
__global__ __launch_bounds__(256) void fri_fold_kernel(const FoldParams p)
{
    constexpr int NX = 1 << RED;
    const u64 M = 1ull << (p.bits - RED);
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= M) return;
    u64 a[3][NX];
    // gather in bit-reversed order so the DIT network below ends in natural order
#pragma unroll
    for (int i = 0; i < NX; i++) {
        int ri = 0;
#pragma unroll
        for (int b = 0; b < RED; b++) ri |= ((i >> b) & 1) << (RED - 1 - b);
        const u64* src = p.pol + 3 * ((u64)ri * M + g);
        a[0][i] = gl::canon(src[0]); a[1][i] = gl::canon(src[1]); a[2][i] = gl::canon(src[2]);
    }
    // inverse NTT, radix-2 DIT, twiddles = powers of w(red)^-1
    u64 wm[RED > 0 ? RED : 1];  // wm[s-1] = root of order 2^s
    {
        u64 w = p.winv_red;
#pragma unroll
        for (int s = RED; s >= 1; s--) { wm[s - 1] = w; w = gl::mulc(w, w); }
    }
#pragma unroll
    for (int s = 1; s <= RED; s++) {
        const int m = 1 << s, h = m >> 1;
        u64 w = 1;
#pragma unroll
        for (int j = 0; j < h; j++) {
#pragma unroll
            for (int k = 0; k < NX; k += m) {
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    u64 t = j ? gl::mulc(w, a[d][k + j + h]) : a[d][k + j + h];
                    u64 u = a[d][k + j];
                    a[d][k + j] = gl::addc(u, t);
                    a[d][k + j + h] = gl::subc(u, t);
                }
            }
            w = gl::mulc(w, wm[s - 1]);
        }
    }
    // coefficient i *= (1/nX) * acc^i, acc = shift_inv * w(bits)^-g ; then Horner at x
    const u64 acc = gl::mulc(p.shift_inv, gl::canon(gl::pow(p.winv_bits, g)));
    u64 r = p.nx_inv;
    e3 c[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) {
#pragma unroll
        for (int d = 0; d < 3; d++) c[i].c[d] = gl::mulc(a[d][i], r);
        r = gl::mulc(r, acc);
    }
    e3 x{{p.x[0], p.x[1], p.x[2]}};
    e3 res = c[NX - 1];
#pragma unroll
    for (int i = NX - 2; i >= 0; i--) res = gl::add3(gl::mul3(res, x), c[i]);
    u64* dst = p.pol2 + 3 * g;
    dst[0] = res.c[0]; dst[1] = res.c[1]; dst[2] = res.c[2];
}

__global__ void copy3_kernel(u64* dst, const u64* src, u64 count)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = gl::canon(src[i]);
}

// dst[i*h + j] = src[j*w + i], ext-3 elements, w = 2^tbits, h = n / w.
// 32x32 element tiles through LDS so both sides move contiguous runs.
__global__ __launch_bounds__(256) void transpose3_kernel(u64* dst, const u64* src, u64 w, u64 h)
{
    __shared__ u64 tile[32][32 * 3 + 1];
    const u64 i0 = (u64)blockIdx.x * 32, j0 = (u64)blockIdx.y * 32;
    for (u32 t = threadIdx.x; t < 32 * 96; t += blockDim.x) {
        u32 jj = t / 96, k = t % 96;          // k = ii*3 + d
        u64 j = j0 + jj, i = i0 + k / 3;
        if (j < h && i < w) tile[jj][k] = src[(j * w + i0) * 3 + k];
    }
    __syncthreads();
    for (u32 t = threadIdx.x; t < 32 * 96; t += blockDim.x) {
        u32 ii = t / 96, k = t % 96;          // k = jj*3 + d
        u32 jj = k / 3, d = k % 3;
        u64 i = i0 + ii, j = j0 + jj;
        if (i < w && j < h) dst[(i * h + j0) * 3 + k] = tile[jj][ii * 3 + d];
    }
}

__device__ __forceinline__ u64 splitmix64(u64 z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// Synthetic trace: x[i] = splitmix64(seed ^ (start+i)) mod p.
__global__ void fill_kernel(u64* dst, u64 n, u64 seed, u64 start)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = gl::canon(splitmix64(seed ^ (start + i)));
}

// Synthetic plookup selector (benchmarks on a synthetic witness only): dst[i][0..dim) = src[(splitmix64(seed ^ (i >> 4)) mod n)][0..dim),
// i.e. an `f` polynomial whose every value occurs in the table `src` -- what a satisfying witness guarantees and a random one
// does not (Polinomial::calculateH1H2 exits on "Number not included", polinomial.hpp:413-414).  Runs of 16 rows select the same
// table row: multiplicities are skewed as in real lookups, and the selector costs 1/16 of a row-by-row random gather.
__global__ void gather_rows_kernel(u64* dst, const u64* src, u64 src_stride, u64 n, u32 dim, u64 seed)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * dim) return;
    const u64 i = t / dim;
    const u32 c = (u32)(t % dim);
    dst[t] = src[(splitmix64(seed ^ (i >> 4)) % n) * src_stride + c];
}
int dev_gather_rows(Ctx* ctx, u64* dst, const u64* src, u64 src_stride, u64 n, u32 dim, u64 seed)
{
    if (!n || !dim) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n * dim + 255) / 256)), dim3(256), 0, ctx->stream, dst, src, src_stride, n, dim, seed);
    XLP_HIP(hipGetLastError());
    return 0;
}

// Field-layer self-test hook (tests/ only): out[i] = op(a[i], b[i]) with the exact device
// functions the kernels use.  op: 0 addc, 1 subc, 2 mulc, 3 canon(a), 4 add(any,canon), 5 sub(any,canon),
// 100+K: canon(mul_pow2c<K>(a)) for the shifts a radix-16 butterfly uses.
__global__ void field_op_kernel(u64* out, const u64* a, const u64* b, u64 n, int op)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 x = a[i], y = b[i];
    u64 r = 0;
    switch (op) {
        case 0: r = gl::addc(gl::canon(x), gl::canon(y)); break;
        case 1: r = gl::subc(gl::canon(x), gl::canon(y)); break;
        case 2: r = gl::mulc(x, y); break;
        case 3: r = gl::canon(x); break;
        case 4: r = gl::canon(gl::add(x, gl::canon(y))); break;
        case 5: r = gl::canon(gl::sub(x, gl::canon(y))); break;
        case 112: r = gl::canon(gl::mul_pow2c<12>(x)); break;
        case 124: r = gl::canon(gl::mul_pow2c<24>(x)); break;
        case 136: r = gl::canon(gl::mul_pow2c<36>(x)); break;
        case 148: r = gl::canon(gl::mul_pow2c<48>(x)); break;
        case 160: r = gl::canon(gl::mul_pow2c<60>(x)); break;
        case 172: r = gl::canon(gl::mul_pow2c<72>(x)); break;
        case 184: r = gl::canon(gl::mul_pow2c<84>(x)); break;
        case 132: r = gl::canon(gl::mul_pow2c<32>(x)); break;
        case 164: r = gl::canon(gl::mul_pow2c<64>(x)); break;
        case 195: r = gl::canon(gl::mul_pow2c<95>(x)); break;
        case 101: r = gl::canon(gl::mul_pow2c<1>(x)); break;
        default: r = gl::canon(gl::mul_pow2(x, (unsigned)(op - 200))); break;   // 200+k: run-time shift k < 96
    }
    out[i] = r;
}

int dev_field_op(Ctx* ctx, u64* out, const u64* a, const u64* b, u64 n, int op)
{
    if (!n) return 0;
    hipLaunchKernelGGL(field_op_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, out, a, b, n, op);
    XLP_HIP(hipGetLastError());
    return 0;
}

template <int RED>
static int launch_fold(Ctx* ctx, const FoldParams& p)
{
    const u64 M = 1ull << (p.bits - RED);
    hipLaunchKernelGGL(fri_fold_kernel<RED>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream, p);
    XLP_HIP(hipGetLastError());
    return 0;
}

int dev_fri_fold(Ctx* ctx, u64* pol2, const u64* pol, u32 bits, u32 red, const u64 x[3], u64 shift_inv)
{
    if (red > bits) return xlp_fail("fri_fold: reduction exceeds polynomial size");
    if (bits > 40) return xlp_fail("fri_fold: polynomial too large");
    if (red == 0) {   // friProve.cpp:87-90: step 0 of a proof whose first step keeps the size
        const u64 cnt = 3ull << bits;
        hipLaunchKernelGGL(copy3_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, pol2, pol, cnt);
        XLP_HIP(hipGetLastError());
        return 0;
    }
    if (red > 6) return xlp_fail("fri_fold: reductionBits > 6 not supported");
    FoldParams p;
    p.pol2 = pol2; p.pol = pol; p.bits = bits; p.red = red;
    for (int d = 0; d < 3; d++) p.x[d] = x[d] % hgl::P;
    p.shift_inv = shift_inv % hgl::P;
    p.winv_bits = hgl::inv(host_w(bits));
    p.winv_red = hgl::inv(host_w(red));
    p.nx_inv = hgl::inv(1ull << red);
    switch (red) {
        case 1: return launch_fold<1>(ctx, p);
        case 2: return launch_fold<2>(ctx, p);
        case 3: return launch_fold<3>(ctx, p);
        case 4: return launch_fold<4>(ctx, p);
        case 5: return launch_fold<5>(ctx, p);
        default: return launch_fold<6>(ctx, p);
    }
}

int dev_transpose3(Ctx* ctx, u64* dst, const u64* src, u64 n, u32 tbits)
{
    if (n == 0) return 0;
    const u64 w = 1ull << tbits;
    if (w > n || n % w) return xlp_fail("transpose: 2^tbits must divide n");
    const u64 h = n / w;
    dim3 grid((unsigned)((w + 31) / 32), (unsigned)((h + 31) / 32));
    if (grid.y > 65535) return xlp_fail("transpose: too many row tiles");
    hipLaunchKernelGGL(transpose3_kernel, grid, dim3(256), 0, ctx->stream, dst, src, w, h);
    XLP_HIP(hipGetLastError());
    return 0;
}

int dev_fill(Ctx* ctx, u64* dst, u64 n, u64 seed, u64 start)
{
    if (!n) return 0;
    u64 blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, dst, n, seed, start);
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
