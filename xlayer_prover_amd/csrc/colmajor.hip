// Column-major device sections (include/xlp_gl.h, "column-major sections"): the reference's slab is row-major (stark_info.cpp:473-482) because
// its loops walk a row at a time; on the GPU a lane is a row, so a section stored as columns is read and written coalesced by every stage of
// Starks::genProof -- the step programs (no staging copy), extendPol (whole lines instead of 64-byte pieces of 6 KB rows), the leaf sponge, the
// FRI polynomial's dot products.  The row-major form survives at the host boundary only: the witness the executor leaves in host memory
// (prover.cpp:435) is transposed band by band as it arrives, the query rows of a proof are gathered at opening time.
//
// Storage order of a column: base domain = natural rows; extended domain = COSET-MAJOR: natural row k = (q << ext) | r sits at storage row
// r * coset_stride + q (coset_stride = rows >> ext unless the caller leaves room for halo rows behind every coset block).
//
// This file: the transposes between the two forms, a strided view copy, and extendPol on column-major sections for the geometries the
// three-kernel fast path (ntt_lde_cm.hip) does not cover -- through a row-major staging chunk and the row-major engine.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gl_field.hpp"
#include "xlp_internal.hpp"

namespace xlp {

// dst[c * cstride + r * cos + q] = canon(src[((q << ext) | r) * ld + col0 + c]),  c < ncols, (q << ext | r) < rows.
// One workgroup: 64 storage rows of one coset x 64 columns through LDS (reads run along the row-major rows, writes along the columns).
__global__ __launch_bounds__(256) void rows_to_cm_kernel(u64* dst, u64 cstride, u64 cos, const u64* src, u64 ld, u64 col0, u32 ncols, u64 nq, u32 ext)
{
    __shared__ u64 tile[64][65];
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // grid: x = row tile * column tiles + column tile (the column tile runs fastest: neighbouring workgroups touch neighbouring bytes of the same
    // rows), z = coset.  (Row tiles used to sit in gridDim.y: 2^17 .. 2^18 of them at 2^23 .. 2^24 rows, past the 65 535 hipDeviceProp.maxGridSize[1] allows.)
    const u32 nct = (ncols + 63) / 64;
    const u32 c0 = (blockIdx.x % nct) * 64;
    const u64 q0 = (u64)(blockIdx.x / nct) * 64;
    const u32 r = blockIdx.z;
    for (u32 i = wv; i < 64; i += 4) {
        const u64 q = q0 + i;
        u64 v = 0;
        if (q < nq && c0 + lane < ncols) v = gl::canon(src[(((q << ext) | r)) * ld + col0 + c0 + lane]);   // what enters a column-major section is canonical (XLP_EXPR_CANONICAL)
        tile[i][lane] = v;
    }
    __syncthreads();
    for (u32 k = wv; k < 64; k += 4) {
        const u64 q = q0 + lane;
        if (c0 + k < ncols && q < nq) dst[(u64)(c0 + k) * cstride + (u64)r * cos + q] = tile[lane][k];
    }
}

__global__ __launch_bounds__(256) void cm_to_rows_kernel(u64* dst, u64 ld, u64 col0, const u64* src, u64 cstride, u64 cos, u32 ncols, u64 nq, u32 ext)
{
    __shared__ u64 tile[64][65];
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // grid: x = row tile * column tiles + column tile (the column tile runs fastest: neighbouring workgroups touch neighbouring bytes of the same
    // rows), z = coset.  (Row tiles used to sit in gridDim.y: 2^17 .. 2^18 of them at 2^23 .. 2^24 rows, past the 65 535 hipDeviceProp.maxGridSize[1] allows.)
    const u32 nct = (ncols + 63) / 64;
    const u32 c0 = (blockIdx.x % nct) * 64;
    const u64 q0 = (u64)(blockIdx.x / nct) * 64;
    const u32 r = blockIdx.z;
    for (u32 k = wv; k < 64; k += 4) {
        const u64 q = q0 + lane;
        u64 v = 0;
        if (c0 + k < ncols && q < nq) v = src[(u64)(c0 + k) * cstride + (u64)r * cos + q];
        tile[lane][k] = v;
    }
    __syncthreads();
    for (u32 i = wv; i < 64; i += 4) {
        const u64 q = q0 + i;
        if (q < nq && c0 + lane < ncols) dst[(((q << ext) | r)) * ld + col0 + c0 + lane] = tile[i][lane];
    }
}

static int check_cm(const char* what, u64 rows, u32 ext, u64& cos)
{
    if (ext > 8) return xlp_fail((std::string(what) + ": more than 2^8 cosets").c_str());
    if (rows & ((1ull << ext) - 1)) return xlp_fail((std::string(what) + ": rows not a multiple of the number of cosets").c_str());
    const u64 nq = rows >> ext;
    if (cos == 0) cos = nq;
    if (cos < nq) return xlp_fail((std::string(what) + ": coset stride smaller than the rows of a coset").c_str());
    return 0;
}

int dev_rows_to_cm(Ctx* ctx, u64* dst, u64 cstride, u64 cos, const u64* src, u64 ld, u64 col0, u64 ncols, u64 rows, u32 ext)
{
    if (!ncols || !rows) return 0;
    if (check_cm("rows -> columns", rows, ext, cos)) return -1;
    if (col0 + ncols > ld) return xlp_fail("rows -> columns: columns exceed the row stride");
    if (ncols >> 31) return xlp_fail("rows -> columns: too many columns");
    const u64 nq = rows >> ext;
    const u64 tiles = ((ncols + 63) / 64) * ((nq + 63) / 64);
    if (tiles >> 31) return xlp_fail("rows <-> columns: more than 2^31 tiles in one launch");
    const dim3 grid((unsigned)tiles, 1, 1u << ext);
    hipLaunchKernelGGL(rows_to_cm_kernel, grid, dim3(256), 0, ctx->stream, dst, cstride, cos, src, ld, col0, (u32)ncols, nq, ext);
    XLP_HIP(hipGetLastError());
    return 0;
}

int dev_cm_to_rows(Ctx* ctx, u64* dst, u64 ld, u64 col0, const u64* src, u64 cstride, u64 cos, u64 ncols, u64 rows, u32 ext)
{
    if (!ncols || !rows) return 0;
    if (check_cm("columns -> rows", rows, ext, cos)) return -1;
    if (col0 + ncols > ld) return xlp_fail("columns -> rows: columns exceed the row stride");
    if (ncols >> 31) return xlp_fail("columns -> rows: too many columns");
    const u64 nq = rows >> ext;
    const u64 tiles = ((ncols + 63) / 64) * ((nq + 63) / 64);
    if (tiles >> 31) return xlp_fail("rows <-> columns: more than 2^31 tiles in one launch");
    const dim3 grid((unsigned)tiles, 1, 1u << ext);
    hipLaunchKernelGGL(cm_to_rows_kernel, grid, dim3(256), 0, ctx->stream, dst, ld, col0, src, cstride, cos, (u32)ncols, nq, ext);
    XLP_HIP(hipGetLastError());
    return 0;
}

// dst[i * drs + c * dcs] = src[i * srs + c * scs]: a Polinomial view (polinomial.hpp:24-40) of either form copied into the other -- what
// surrounds the row-scale helpers that want dense ext-3 elements (calculateZ, calculateH1H2) when the sections are column-major
__global__ __launch_bounds__(256) void copy_view_kernel(u64* dst, u64 drs, u64 dcs, const u64* src, u64 srs, u64 scs, u64 n, u32 dim)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (u32 c = 0; c < dim; c++) dst[i * drs + c * dcs] = src[i * srs + c * scs];
}
int dev_copy_view(Ctx* ctx, u64* dst, u64 drs, u64 dcs, const u64* src, u64 srs, u64 scs, u64 n, u32 dim)
{
    if (!n || !dim) return 0;
    hipLaunchKernelGGL(copy_view_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dst, drs, dcs, src, srs, scs, n, dim);
    XLP_HIP(hipGetLastError());
    return 0;
}

// dst[a * ds0 + b * ds1 + i] = src[a * ss0 + b * ss1 + i], a < n0, b < n1, i < n: runs of n contiguous words regrouped -- the packing and
// unpacking around the exchanges of a row-sharded proof on column-major windows (a rank's rows of a column, of a coset, are one run), the
// halo rows of the windows (n = 1), the windows of the replicated constants
__global__ __launch_bounds__(256) void copy_3d_kernel(u64* dst, u64 ds0, u64 ds1, const u64* src, u64 ss0, u64 ss1, u64 n1, u64 n, u64 chunks)
{
    const u64 blk = blockIdx.x;
    const u64 run = blk / chunks, ch = blk % chunks;
    const u64 a = run / n1, b = run % n1;
    const u64 i = ch * 1024 + threadIdx.x;
    const u64* s = src + a * ss0 + b * ss1;
    u64* d = dst + a * ds0 + b * ds1;
#pragma unroll
    for (int k = 0; k < 4; k++) if (i + k * 256 < n) d[i + k * 256] = s[i + k * 256];
}
// short runs: one thread per word
__global__ __launch_bounds__(256) void copy_3d_small_kernel(u64* dst, u64 ds0, u64 ds1, const u64* src, u64 ss0, u64 ss1, u64 n1, u64 n, u64 total)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const u64 i = t % n, run = t / n, a = run / n1, b = run % n1;
    dst[a * ds0 + b * ds1 + i] = src[a * ss0 + b * ss1 + i];
}
int dev_copy_3d(Ctx* ctx, u64* dst, u64 ds0, u64 ds1, const u64* src, u64 ss0, u64 ss1, u64 n0, u64 n1, u64 n)
{
    if (!n0 || !n1 || !n) return 0;
    if (n < 256) {
        const u64 total = n0 * n1 * n;
        hipLaunchKernelGGL(copy_3d_small_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, dst, ds0, ds1, src, ss0, ss1, n1, n, total);
    } else {
        const u64 chunks = (n + 1023) / 1024, blocks = n0 * n1 * chunks;
        if (blocks >> 31) return xlp_fail("copy 3d: too many blocks");
        hipLaunchKernelGGL(copy_3d_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, dst, ds0, ds1, src, ss0, ss1, n1, n, chunks);
    }
    XLP_HIP(hipGetLastError());
    return 0;
}

// extendPol (starks.cpp:53,134,215) on column-major sections: in = ncols columns of n rows (in_cstride apart), out = ncols columns of n_ext rows,
// coset-major (out_cstride apart, cosets out_cos apart).  2^17 <= n <= 2^23 with blow-up <= 8 takes the three column-major kernels of
// ntt_lde_cm.hip; every other geometry goes through a row-major staging chunk of <= 64 columns and the row-major engine.
int dev_extend_cm_fast(Ctx* ctx, u64* out, u64 out_cstride, u64 out_cos, const u64* in, u64 in_cstride, u32 L, u32 e, u64 ncols);   // ntt_lde_cm.hip
bool lde_cm_fast_supported(u32 L, u32 e);

int dev_extend_cm(Ctx* ctx, u64* out, u64 out_cstride, u64 out_cos, const u64* in, u64 in_cstride, u64 n_ext, u64 n, u64 ncols)
{
    if (n == 0 || ncols == 0) return 0;
    if ((n & (n - 1)) || (n_ext & (n_ext - 1)) || n_ext < n) return xlp_fail("extendPol (column-major): sizes must be powers of two, n_ext >= n");
    u32 L = 0, Le = 0;
    while ((1ull << L) < n) L++;
    while ((1ull << Le) < n_ext) Le++;
    if (Le > 32) return xlp_fail("extendPol (column-major): size exceeds 2^32");
    const u32 e = Le - L;
    if (out_cos == 0) out_cos = n;
    if (out_cos < n) return xlp_fail("extendPol (column-major): coset stride smaller than the base domain");
    static const bool no_fast = getenv("XLP_LDE_CM_FAST") && atoi(getenv("XLP_LDE_CM_FAST")) == 0;
    if (!no_fast && lde_cm_fast_supported(L, e)) return dev_extend_cm_fast(ctx, out, out_cstride, out_cos, in, in_cstride, L, e, ncols);
    const u64 wc_max = ncols < 64 ? ncols : 64;
    u64* stage = nullptr;
    if (ctx_scratch(ctx, (size_t)(n + n_ext) * wc_max * sizeof(u64), 6, &stage)) return -1;
    u64* stage_out = stage + n * wc_max;
    for (u64 c0 = 0; c0 < ncols; c0 += wc_max) {
        const u64 wc = ncols - c0 < wc_max ? ncols - c0 : wc_max;
        if (dev_cm_to_rows(ctx, stage, wc, 0, in + c0 * in_cstride, in_cstride, n, wc, n, 0)) return -1;
        if (dev_extend(ctx, stage_out, stage, n_ext, n, wc, nullptr)) return -1;
        if (dev_rows_to_cm(ctx, out + c0 * out_cstride, out_cstride, out_cos, stage_out, wc, 0, wc, n_ext, e)) return -1;
    }
    return 0;
}

}  // namespace xlp
