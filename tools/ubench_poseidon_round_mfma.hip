// Microbenchmark (round 5, a lead for round 6): one FULL ROUND of Poseidon-Goldilocks (12 S-boxes x^7, then st <- M st + c) with the state of a
// permutation in ONE lane, as csrc/poseidon.hip keeps it, and the linear layer either
//   VALU  as in csrc/poseidon.hip mds<true>: 288 multiply-adds against the small circulant constants + 12 recombinations, or
//   MFMA  on the matrix pipe in the form of csrc/ntt_lde_mid_mfma.hip: the two halves of the wave trade six words each (v_permlane32_swap: lanes 0-31 keep
//         words {0,1,4,5,8,9} of their own state and receive the same words of the state 32 lanes up; lanes 32-63 the other six), which turns the 64 states
//         into two groups of 32 in the MFMA operand layout; v_mfma_i32_32x32x32_i8 with the 12 x 12 matrix + a 13th column that carries the round constant
//         (input word 12 is the constant 1; the column also absorbs the xor-0x80 offset), 12 MFMA per group; recombine(); trade back.
// Both must agree bit for bit after canonicalisation.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I xlayer_prover_amd/csrc -I include tools/ubench_poseidon_round_mfma.hip -o build/ubench_poseidon_round_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "ntt_mfma_common.hpp"
namespace xlp { bool g_plan_active = false; }
using gl::u32;
using gl::u64;
using xlp::v4i;
using xlp::v16i;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ u64 pow7(u64 a)
{
    u64 a2 = gl::sqr(a), a4 = gl::sqr(a2), a3 = gl::mul(a, a2);
    return gl::mul(a3, a4);
}
__device__ __forceinline__ u64 recombine2(u64 L, u64 H)      // csrc/poseidon.hip recombine
{
    const u32 Hl = (u32)H, Hh = (u32)(H >> 32);
    const u64 t = (u64)Hh * 0xFFFFFFFFu + L;
    u32 c1, c2, c3;
    u32 rhi = __builtin_addc((u32)(t >> 32), Hl, 0u, &c1);
    u32 rlo = (u32)t;
    gl::opaque(rhi);
    const u32 fix = c1 ? 0xFFFFFFFFu : 0u;
    rlo = __builtin_addc(rlo, fix, 0u, &c2);
    rhi = __builtin_addc(rhi, 0u, c2, &c3);
    gl::opaque(rlo); gl::opaque(rhi);
    return gl::pack(rlo, rhi);
}
struct MdsConsts { u32 mc[13]; };
__device__ __forceinline__ MdsConsts mds_consts()
{
    constexpr u32 MC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    MdsConsts k;
#pragma unroll
    for (int i = 0; i < 12; i++) { k.mc[i] = MC[i]; asm volatile("" : "+s"(k.mc[i])); }
    k.mc[12] = MC[0] + 8; asm volatile("" : "+s"(k.mc[12]));
    return k;
}
__device__ __forceinline__ void mds_valu(u64 (&st)[12], const u64* __restrict__ c, const MdsConsts& k)
{
    u32 lo[12], hi[12];
#pragma unroll
    for (int j = 0; j < 12; j++) { lo[j] = (u32)st[j]; hi[j] = (u32)(st[j] >> 32); }
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u64 L = (u32)c[i], H = c[i] >> 32;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const u32 m = (i == 0 && j == 0) ? k.mc[12] : k.mc[(j - i + 12) % 12];
            L += (u64)lo[j] * m;
            H += (u64)hi[j] * m;
        }
        st[i] = recombine2(L, H);
    }
}

__constant__ u64 RC[12];

__global__ __launch_bounds__(256) void round_valu(u64* state, u64 nperm, int iters)
{
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nperm) return;
    const MdsConsts k = mds_consts();
    u64 st[12];
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = state[12 * p + i];
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = pow7(st[i]);
        mds_valu(st, RC, k);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) state[12 * p + i] = gl::canon(st[i]);
}

// swap the upper half of a (lanes 32..63) with the lower half of b (lanes 0..31)
__device__ __forceinline__ void half_swap(u64& a, u64& b)
{
    u32 al = (u32)a, ah = (u32)(a >> 32), bl = (u32)b, bh = (u32)(b >> 32);
    auto r0 = __builtin_amdgcn_permlane32_swap(al, bl, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
    a = gl::pack(r0[0], r1[0]); b = gl::pack(r0[1], r1[1]);
}

__global__ __launch_bounds__(256) void round_mfma(u64* state, u64 nperm, int iters, const v4i* Atab, const int* bias)
{
    using namespace xlp;
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nperm) return;               // (nperm is a multiple of 256 here: no partial wave)
    const u32 lane = threadIdx.x & 63;
    v4i A[3][4];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) A[t][q] = Atab[(t * 4 + q) * 64 + lane];
    v16i C;
#pragma unroll
    for (int i = 0; i < 16; i++) C[i] = bias[i & 7];
    const RecK K = rec_consts();
    // input word 12 = the constant 1 (lanes h = 0, slot q = 3, ee = 0), xor-ed already; the other three words of that chunk are 0 (xor-ed: 0x80..80)
    v4i b3;
    {
        const u32 h = lane >> 5;
        const u64 one = h == 0 ? 1ull : 0ull;
        b3.x = (int)((u32)one ^ 0x80808080u); b3.y = (int)0x80808080u; b3.z = (int)0x80808080u; b3.w = (int)0x80808080u;
    }
    u64 st[12];
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = state[12 * p + i];
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = pow7(st[i]);
        // trade: words a = 4q + e stay / words b = a + 2 travel
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int e = 0; e < 2; e++) half_swap(st[4 * q + e], st[4 * q + 2 + e]);
#pragma unroll
        for (int g = 0; g < 2; g++) {         // group 0: words 4q + e (as traded), group 1: words 4q + 2 + e
            v4i b[3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const u64 w0 = st[4 * q + 2 * g], w1 = st[4 * q + 2 * g + 1];
                b[q].x = (int)((u32)w0 ^ 0x80808080u); b[q].y = (int)((u32)(w0 >> 32) ^ 0x80808080u);
                b[q].z = (int)((u32)w1 ^ 0x80808080u); b[q].w = (int)((u32)(w1 >> 32) ^ 0x80808080u);
            }
#pragma unroll
            for (int t = 0; t < 3; t++) {
                v16i acc = C;
#pragma unroll
                for (int q = 0; q < 3; q++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t][q], b[q], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t][3], b3, acc, 0, 0, 0);
                int d[16];
#pragma unroll
                for (int i = 0; i < 16; i++) d[i] = acc[i];
                st[4 * t + 2 * g] = recombine(d, K, 0u, 0u);
                st[4 * t + 2 * g + 1] = recombine(d + 8, K, 0u, 0u);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int e = 0; e < 2; e++) half_swap(st[4 * q + e], st[4 * q + 2 + e]);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) state[12 * p + i] = gl::canon(st[i]);
}

int main(int argc, char** argv)
{
    const u64 nperm = 1ull << (argc > 1 ? atoi(argv[1]) : 22);
    const int iters = argc > 2 ? atoi(argv[2]) : 16;
    std::vector<u64> h(12 * nperm);
    u64 s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; }
    u64 rc[12];
    for (int i = 0; i < 12; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; rc[i] = s % xlp::hgl::P; }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(RC), rc, sizeof rc));
    u64 *da, *db; int* dbias; v4i* dA;
    CK(hipMalloc(&da, h.size() * 8)); CK(hipMalloc(&db, h.size() * 8));
    {
        using namespace xlp;
        constexpr u32 MC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
        u64 M[16][16] = {};
        const u64 B0 = 0x8080808080808080ull % hgl::P;
        const u64 inv1mb0 = hgl::inv(hgl::sub(1, B0));
        for (int i = 0; i < 12; i++) {
            u64 rs = 0;
            for (int j = 0; j < 12; j++) { M[i][j] = MC[((j - i) % 12 + 12) % 12] + ((i == 0 && j == 0) ? 8u : 0u); rs += M[i][j]; }
            // column 12 times (1 - B0) must give c_i + B0 * rowsum_i: the MFMA sees every word minus B0, the word "1" included
            M[i][12] = hgl::mul(hgl::add(rc[i], hgl::mul(B0, rs)), inv1mb0);
        }
        // the three unused words of chunk 3 (13, 14, 15) are 0 and see -B0: their columns are zero, so they add nothing
        std::vector<uint8_t> img(16384); emit_matrix(M, img.data());
        int bias[8]; bias_digits(0, bias);
        CK(hipMalloc(&dA, 16384)); CK(hipMemcpy(dA, img.data(), 16384, hipMemcpyHostToDevice));
        CK(hipMalloc(&dbias, sizeof bias)); CK(hipMemcpy(dbias, bias, sizeof bias, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms[2] = {0, 0};
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemcpy(da, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(round_valu, dim3((unsigned)(nperm / 256)), dim3(256), 0, 0, da, nperm, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[0], e0, e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(round_mfma, dim3((unsigned)(nperm / 256)), dim3(256), 0, 0, db, nperm, iters, dA, dbias);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[1], e0, e1));
    }
    std::vector<u64> ra(h.size()), rb(h.size());
    CK(hipMemcpy(ra.data(), da, h.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(rb.data(), db, h.size() * 8, hipMemcpyDeviceToHost));
    u64 bad = 0;
    for (size_t i = 0; i < h.size(); i++) bad += ra[i] != rb[i];
    const double n = (double)nperm * iters;
    printf("full round x %d on 2^%d states (12 S-boxes + linear layer with round constants): VALU %.3f ms (%.2f G rounds/s)   MFMA linear layer %.3f ms (%.2f G rounds/s)   ratio %.2f   mismatches %llu\n",
           iters, (int)__builtin_ctzll(nperm), ms[0], n / ms[0] / 1e6, ms[1], n / ms[1] / 1e6, ms[0] / ms[1], (unsigned long long)bad);
    return bad != 0;
}
