// Microbenchmark for the question BASELINE.json's north star raises: is the Poseidon 12x12 MDS mat-vec better on the
// MFMA pipe?  Two kernels apply the MDS (no S-box) `iters` times to the same states and must agree:
//   A  the production form (csrc/poseidon.hip mds<>): one permutation per lane, 288 v_mad_u64_u32 per MDS
//   B  v_mfma_i32_16x16x32_i8: 16 permutations per wave, 3 state words per lane; the state is split into 8 byte limbs
//      (xor 0x80 -> signed), one MFMA per limb against the constant matrix, the eight int32 limb sums of each word are
//      recombined (6 multiply-adds) and reduced as in A.  The row permutation of A's operand makes the MFMA's output
//      layout equal its input layout, so no cross-lane traffic is needed.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I xlayer_prover_amd/csrc -I include tools/ubench_mds_mfma.hip -o build/ubench_mds_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "ntt_mfma_common.hpp"
namespace xlp { bool g_plan_active = false; }      // (xlp_internal.hpp routes kernel launches through the plan-mode switch of the library)
using gl::u32;
using gl::u64;
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mds_valu(u64 (&st)[12])
{
    constexpr u32 MC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 lo[12], hi[12];
#pragma unroll
    for (int j = 0; j < 12; j++) { lo[j] = (u32)st[j]; hi[j] = (u32)(st[j] >> 32); }
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u64 L = 0, H = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const u32 m = MC[(j - i + 12) % 12] + ((i == 0 && j == 0) ? 8u : 0u);
            L += (u64)lo[j] * m;
            H += (u64)hi[j] * m;
        }
        const u64 Hl = (u64)(u32)H << 32, Hh = H >> 32;
        u64 lo64 = L + Hl;
        const u64 cy = lo64 < Hl;
        lo64 += (Hh + cy) * gl::EPS;
        if (lo64 < (Hh + cy) * gl::EPS) lo64 += gl::EPS;
        st[i] = lo64;
    }
}

__global__ __launch_bounds__(256) void mds_a(u64* state, u64 nperm, int iters)
{
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nperm) return;
    u64 st[12];
#pragma unroll
    for (int k = 0; k < 12; k++) st[k] = state[12 * p + k];
    for (int it = 0; it < iters; it++) mds_valu(st);
#pragma unroll
    for (int k = 0; k < 12; k++) state[12 * p + k] = gl::canon(st[k]);
}

__device__ __forceinline__ u32 mcoef(int o, int e)
{
    constexpr u32 MC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    return MC[((e - o) % 12 + 12) % 12] + ((o == 0 && e == 0) ? 8u : 0u);
}

__global__ __launch_bounds__(256) void mds_b(u64* state, u64 nperm, int iters)
{
    const u32 lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u64 p = wave * 16 + j;
    if (wave * 16 >= nperm) return;
    const u64 pp = p < nperm ? p : nperm - 1;
    u64 e[3];
#pragma unroll
    for (int r = 0; r < 3; r++) e[r] = state[12 * pp + 3 * g + r];
    // A operand: row i = lane % 16 -> output word o = 3 (i / 4) + i % 4 (rows with i % 4 == 3 unused);
    //            k = 8 (lane / 16) + b -> input word 3 (lane / 16) + b, b < 3
    u64 A = 0;
    {
        const int i = lane & 15, kg = lane >> 4;
        if ((i & 3) != 3) {
            const int o = 3 * (i >> 2) + (i & 3);
            for (int b = 0; b < 3; b++) A |= (u64)mcoef(o, 3 * kg + b) << (8 * b);
        }
    }
    // C: + 128 * rowsum undoes the xor 0x80 of the inputs
    v4i C;
    for (int r = 0; r < 4; r++) {
        int s = 0;
        if (r < 3) for (int x = 0; x < 12; x++) s += (int)mcoef(3 * g + r, x);
        C[r] = 128 * s;
    }
    for (int it = 0; it < iters; it++) {
        u32 L[3], H[3];
#pragma unroll
        for (int r = 0; r < 3; r++) { L[r] = (u32)e[r]; H[r] = (u32)(e[r] >> 32); }
        v4i D[8];
#pragma unroll
        for (int l = 0; l < 8; l++) {
            const u32* S = l < 4 ? L : H;
            const u32 q = l & 3;
            const u32 t = __builtin_amdgcn_perm(S[1], S[0], 0x0c0c0000u | ((4 + q) << 8) | q);
            u32 b = __builtin_amdgcn_perm(S[2], t, 0x0c000100u | ((4 + q) << 16));
            b ^= 0x00808080u;
            D[l] = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)A, (long)(u64)b, C, 0, 0, 0);   // D[out row][perm col]
        }
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const u64 X = (u64)(u32)D[0][r] + ((u64)(u32)D[1][r] << 8) + ((u64)(u32)D[2][r] << 16) + ((u64)(u32)D[3][r] << 24);
            const u64 Y = (u64)(u32)D[4][r] + ((u64)(u32)D[5][r] << 8) + ((u64)(u32)D[6][r] << 16) + ((u64)(u32)D[7][r] << 24);
            const u64 Hl = (u64)(u32)Y << 32, Hh = Y >> 32;
            u64 lo64 = X + Hl;
            const u64 cy = lo64 < Hl;
            lo64 += (Hh + cy) * gl::EPS;
            if (lo64 < (Hh + cy) * gl::EPS) lo64 += gl::EPS;
            e[r] = lo64;
        }
    }
    if (p < nperm)
#pragma unroll
        for (int r = 0; r < 3; r++) state[12 * p + 3 * g + r] = gl::canon(e[r]);
}

// C (round 5): v_mfma_i32_32x32x32_i8, 32 permutations per wave, a lane holds 6 of the 12 state words.  A word is its own eight byte limbs (xor 0x80), the
// 12 x 12 matrix is emitted like the DFT matrices of csrc/ntt_lde_mid_mfma.hip (ntt_mfma_common.hpp emit_matrix: rows / columns >= 12 zero), the eight digit
// sums of an output word are recombined by recombine() (13 vector instructions); the xor's offset B0 * rowsum(k) enters like a round constant would
// (two 64-bit adds inside recombine): no byte permutes, 9 MFMA per 32 MDS.
__global__ __launch_bounds__(256) void mds_c(u64* state, u64 nperm, int iters, const xlp::v4i* Atab, const int* bias, const u64* k0)
{
    using namespace xlp;
    const u32 lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u64 p = wave * 32 + n;
    if (wave * 32 >= nperm) return;
    const u64 pp = p < nperm ? p : nperm - 1;
    v4i A[3][3];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int q = 0; q < 3; q++) A[t][q] = Atab[(t * 4 + q) * 64 + lane];
    v16i C;
#pragma unroll
    for (int i = 0; i < 16; i++) C[i] = bias[i & 7];
    const RecK K = rec_consts();
    u64 e[6], kc[6];                      // words j = 4q + 2h + ee, q < 3
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int ee = 0; ee < 2; ee++) { e[2 * q + ee] = state[12 * pp + 4 * q + 2 * h + ee]; kc[2 * q + ee] = k0[4 * q + 2 * h + ee]; }
    for (int it = 0; it < iters; it++) {
        v4i b[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            b[q].x = (int)((u32)e[2 * q] ^ 0x80808080u); b[q].y = (int)((u32)(e[2 * q] >> 32) ^ 0x80808080u);
            b[q].z = (int)((u32)e[2 * q + 1] ^ 0x80808080u); b[q].w = (int)((u32)(e[2 * q + 1] >> 32) ^ 0x80808080u);
        }
#pragma unroll
        for (int t = 0; t < 3; t++) {
            v16i acc = C;
#pragma unroll
            for (int q = 0; q < 3; q++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t][q], b[q], acc, 0, 0, 0);
            int d[16];
#pragma unroll
            for (int i = 0; i < 16; i++) d[i] = acc[i];
            e[2 * t] = recombine(d, K, (u32)kc[2 * t], (u32)(kc[2 * t] >> 32));
            e[2 * t + 1] = recombine(d + 8, K, (u32)kc[2 * t + 1], (u32)(kc[2 * t + 1] >> 32));
        }
    }
    if (p < nperm)
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int ee = 0; ee < 2; ee++) state[12 * p + 4 * q + 2 * h + ee] = gl::canon(e[2 * q + ee]);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const u64 nperm = 1ull << (argc > 1 ? atoi(argv[1]) : 22);
    const int iters = argc > 2 ? atoi(argv[2]) : 64;
    std::vector<u64> h(12 * nperm);
    u64 s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; }
    u64 *da, *db, *dc, *dk0; int* dbias; xlp::v4i* dA;
    CK(hipMalloc(&da, h.size() * 8)); CK(hipMalloc(&db, h.size() * 8)); CK(hipMalloc(&dc, h.size() * 8));
    {
        constexpr u32 MC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
        u64 M[16][16] = {}; u64 k0h[16] = {};
        const u64 B0 = 0x8080808080808080ull % xlp::hgl::P;
        for (int i = 0; i < 12; i++) { u64 rs = 0; for (int j = 0; j < 12; j++) { M[i][j] = MC[((j - i) % 12 + 12) % 12] + ((i == 0 && j == 0) ? 8u : 0u); rs += M[i][j]; } k0h[i] = xlp::hgl::mul(B0, rs); }
        std::vector<uint8_t> img(16384); xlp::emit_matrix(M, img.data());
        int bias[8]; xlp::bias_digits(0, bias);
        CK(hipMalloc(&dA, 16384)); CK(hipMemcpy(dA, img.data(), 16384, hipMemcpyHostToDevice));
        CK(hipMalloc(&dbias, sizeof bias)); CK(hipMemcpy(dbias, bias, sizeof bias, hipMemcpyHostToDevice));
        CK(hipMalloc(&dk0, sizeof k0h)); CK(hipMemcpy(dk0, k0h, sizeof k0h, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms[3] = {0, 0, 0};
    for (int rep = 0; rep < 2; rep++) {        // rep 0 warms up
        CK(hipMemcpy(da, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dc, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mds_c, dim3((unsigned)((nperm * 2 + 255) / 256)), dim3(256), 0, 0, dc, nperm, iters, dA, dbias, dk0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[2], e0, e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mds_a, dim3((unsigned)((nperm + 255) / 256)), dim3(256), 0, 0, da, nperm, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[0], e0, e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mds_b, dim3((unsigned)((nperm * 4 + 255) / 256)), dim3(256), 0, 0, db, nperm, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[1], e0, e1));
    }
    std::vector<u64> ra(h.size()), rb(h.size()), rc(h.size());
    CK(hipMemcpy(rc.data(), dc, h.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ra.data(), da, h.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(rb.data(), db, h.size() * 8, hipMemcpyDeviceToHost));
    u64 bad = 0, badc = 0;
    for (size_t i = 0; i < h.size(); i++) { bad += ra[i] != rb[i]; badc += ra[i] != rc[i]; }
    const double n = (double)nperm * iters;
    printf("MDS x %d on 2^%d states: VALU %.3f ms (%.2f G MDS/s)   MFMA-i8 %.3f ms (%.2f G MDS/s)   ratio %.2f   mismatches %llu\n", iters,
           (int)__builtin_ctzll(nperm), ms[0], n / ms[0] / 1e6, ms[1], n / ms[1] / 1e6, ms[0] / ms[1], (unsigned long long)bad);
    printf("round 5 form (32x32x32, words as their own limbs, 13-instruction recombination): %.3f ms (%.2f G MDS/s)   ratio to VALU %.2f   mismatches %llu\n", ms[2], n / ms[2] / 1e6, ms[0] / ms[2], (unsigned long long)badc);
    return bad != 0 || badc != 0;
}
