// Poseidon-Goldilocks permutation, linear hash and Merkle tree for gfx950.
//
// Replaces PoseidonGoldilocks::hash_full_result (transcript.cpp:23,46),
// ::merkletree_avx / merkletree_avx512 (merkleTreeGL.cpp:39-43) of the
// reference's absent goldilocks submodule.  Permutation parameters restated
// from src/sm/poseidon_g/poseidon_g_executor.cpp:201-230 (t=12, RF=8, RP=22,
// x^7, MDS = circ(17,15,41,16,2,28,13,13,39,18,34,20) + diag(8,0,...)).
//
// The kernels are integer-ALU bound: one permutation per lane,
// state in 24 VGPRs, MDS by 32x32->64 multiply-adds against the small
// circulant constants (the two 32-bit halves of each state word are
// accumulated separately so no product exceeds 64 bits), one reduction per
// output word.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gl_field.hpp"
#include "xlp_internal.hpp"

namespace xlp {
using gl::u32;
using gl::u64;

__constant__ u64 POSEIDON_C[360] = {
#include "poseidon_constants.inc"
};
#include "poseidon_fused.inc"

__device__ __forceinline__ u64 pow7(u64 a)
{
    u64 a2 = gl::sqr(a), a4 = gl::sqr(a2), a3 = gl::mul(a, a2);
    return gl::mul(a3, a4);
}

// The circulant constants, laundered into SGPRs once per kernel: as literals the compiler strength-reduces products like
// x * 17 into shift-add pairs (v_lshl_add_u64, half rate like the multiply-add they replace, but two of them) -- 364
// operations for the 288 multiply-adds of one MDS.  mc[12] = M[0][0] = 17 + 8.
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

// L + H * 2^32 (mod p) -> any, for L, H < 2^57 (the sums of the multiply-add chains below).
// value = L + Hh * 2^64 + Hl * 2^32 = (L + Hh * EPS) + Hl * 2^32 (mod p), Hh < 2^25: one multiply-add (no overflow: < 2^58), one
// 32-bit add into the high word, and + EPS if that carried (the wrapped value is < 2^58, so the correction cannot carry again).
__device__ __forceinline__ u64 recombine(u64 L, u64 H)
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

// st <- M * st + c, M[i][j] = MCIRC[(j-i) mod 12] (+8 on [0][0]); inputs "any", outputs "any".
// `c` (12 canonical words, wave-uniform -> SGPRs) is the NEXT round's constant vector: it is the
// initial value of the multiply-add chains, so the per-round constant addition costs no instruction.
template <bool ADDC>
__device__ __forceinline__ void mds(u64 (&st)[12], const u64* __restrict__ c, const MdsConsts& k)
{
    u32 lo[12], hi[12];
#pragma unroll
    for (int j = 0; j < 12; j++) { lo[j] = (u32)st[j]; hi[j] = (u32)(st[j] >> 32); }
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u64 L = 0, H = 0;
        if constexpr (ADDC) { L = (u32)c[i]; H = c[i] >> 32; }
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const u32 m = (i == 0 && j == 0) ? k.mc[12] : k.mc[(j - i + 12) % 12];
            L += (u64)lo[j] * m;   // < 2^32 * 264 + 2^32 < 2^41
            H += (u64)hi[j] * m;
        }
        st[i] = recombine(L, H);
    }
}

// Three partial rounds as ONE linear layer (tools/gen_poseidon_fused.py has the algebra).  Only word 0 goes through the S-box, so
// with Z = diag(0, 1, ..., 1) the state after three rounds is (M Z M Z M) u' + (M Z M e0) y2 + (M e0) y3 + K, where u' = u with
// word 0 replaced by y1 = u[0]^7 and y2, y3 are the next two S-box outputs, themselves x^7 of one 12-term (13-, 14-term) dot
// product with row 0 of M (of M Z M).  The integer matrices stay below 2^22, so every product against a 32-bit half of a word
// still fits the 64-bit multiply-add chains: 14 x 12 x 2 + 54 multiply-adds and 14 recombinations for three rounds instead of
// 3 x (288 + 12) and 36 -- 20 % fewer instructions per permutation, and the permutation IS the Merkle commit (integer-issue bound).
// Field arithmetic is exact, so the result is the same element as round-by-round evaluation (every known answer still holds).
__constant__ u32 POSEIDON_FU32[193] = XLP_POSEIDON_FUSED_U32;
__constant__ u64 POSEIDON_FU64[98] = XLP_POSEIDON_FUSED_U64;

__device__ __forceinline__ void partial3(u64 (&st)[12], const u64* __restrict__ kc)
{
    // The table pointer is laundered per call: as a loop invariant the compiler loads all 193 constants ahead of the loop over the
    // groups, runs out of SGPRs and parks them in VGPR lanes (311 v_readlane per call, VALU slots); loaded where they are used they
    // are scalar loads beside the VALU stream.
    // (the pointer keeps the CONSTANT address space through the SGPR laundering below: typed as a generic pointer it lost it, and the 193 constants
    // arrived by 49 flat_load_dwordx4 of a uniform address into VECTOR registers per call, 7 calls per permutation; now 77 s_load, 17 registers fewer.
    // Round 5 A/B on one box, profiles/r5/poseidon_scalar_constants_ab.json: leaf sponge 2.712 -> 2.857 G permutations/s, resident ceiling 2.765 -> 2.860)
    typedef const __attribute__((address_space(4))) u32* cu32p;
    cu32p F = (cu32p)POSEIDON_FU32;
    asm volatile("" : "+s"(F));
    cu32p R1 = F;             // row 0 of M
    cu32p R2 = F + 12;        // row 0 of M Z M
    const u32 m00 = F[24];
    cu32p B3 = F + 25;        // M Z M Z M, row-major
    cu32p B2 = F + 25 + 144;  // M Z M e0
    cu32p A1 = F + 25 + 156;  // M e0
    st[0] = pow7(st[0]);
    u32 lo[12], hi[12];
#pragma unroll
    for (int j = 0; j < 12; j++) { lo[j] = (u32)st[j]; hi[j] = (u32)(st[j] >> 32); }
    u64 L = (u32)kc[0], H = kc[0] >> 32;
#pragma unroll
    for (int j = 0; j < 12; j++) { L += (u64)lo[j] * R1[j]; H += (u64)hi[j] * R1[j]; }
    const u64 y2 = pow7(recombine(L, H));
    const u32 y2l = (u32)y2, y2h = (u32)(y2 >> 32);
    L = (u32)kc[1]; H = kc[1] >> 32;
#pragma unroll
    for (int j = 0; j < 12; j++) { L += (u64)lo[j] * R2[j]; H += (u64)hi[j] * R2[j]; }
    L += (u64)y2l * m00; H += (u64)y2h * m00;
    const u64 y3 = pow7(recombine(L, H));
    const u32 y3l = (u32)y3, y3h = (u32)(y3 >> 32);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        L = (u32)kc[2 + i]; H = kc[2 + i] >> 32;
#pragma unroll
        for (int j = 0; j < 12; j++) { L += (u64)lo[j] * B3[12 * i + j]; H += (u64)hi[j] * B3[12 * i + j]; }
        L += (u64)y2l * B2[i]; H += (u64)y2h * B2[i];
        L += (u64)y3l * A1[i]; H += (u64)y3h * A1[i];
        st[i] = recombine(L, H);
    }
}

// Round r: state already holds + C[r] (added by the previous round's linear layer).  Rounds 0-3 and 26-29 are full, 4-25 partial:
// 4-24 run as seven fused triples, 25 as a plain partial round.
__device__ __forceinline__ void permute(u64 (&st)[12])
{
    const MdsConsts k = mds_consts();
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = gl::add(st[i], POSEIDON_C[i]);  // any + canon
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = pow7(st[i]);
        mds<true>(st, &POSEIDON_C[12 * (r + 1)], k);
    }
#pragma unroll 1
    for (int g = 0; g < 7; g++) partial3(st, &POSEIDON_FU64[14 * g]);
    st[0] = pow7(st[0]);
    mds<true>(st, &POSEIDON_C[12 * 26], k);
#pragma unroll 1
    for (int r = 26; r < 29; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = pow7(st[i]);
        mds<true>(st, &POSEIDON_C[12 * (r + 1)], k);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = pow7(st[i]);
    mds<false>(st, nullptr, k);
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = gl::canon(st[i]);
}

// hash_full_result on `count` independent 12-word states.
__global__ __launch_bounds__(256) void perm_kernel(u64* out, const u64* in, u64 count)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 st[12];
#pragma unroll
    for (int k = 0; k < 12; k++) st[k] = in[12 * i + k];
    permute(st);
#pragma unroll
    for (int k = 0; k < 12; k++) out[12 * i + k] = st[k];
}

// MEASUREMENT AID ("perms/s vs an int-ALU ceiling measured by a register-resident microbenchmark"): the leaf sponge's
// arithmetic with no memory traffic at all -- every lane walks `iters` chained permutations of a state that never leaves its registers
// (the capacity is fed back as in the sponge, the rate words are rotated out of the previous output), launched with the sponge's own
// launch bounds and grid.  What this kernel reaches is the ceiling of THIS formulation of the permutation on this chip; the sponge's
// distance from it is what its loads cost.
__global__ __launch_bounds__(256) void perm_resident_kernel(u64* out, u64 count, u64 iters)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 st[12];
#pragma unroll
    for (int k = 0; k < 12; k++) st[k] = (i + 1) * 0x9E3779B97F4A7C15ULL + (u64)k * 0xBF58476D1CE4E5B9ULL;
#pragma unroll 1
    for (u64 it = 0; it < iters; it++) {
        permute(st);
        const u64 c0 = st[0], c1 = st[1], c2 = st[2], c3 = st[3];
        st[0] = st[4]; st[1] = st[5]; st[2] = st[6]; st[3] = st[7]; st[4] = st[8]; st[5] = st[9]; st[6] = st[10]; st[7] = st[11];
        st[8] = c0; st[9] = c1; st[10] = c2; st[11] = c3;
    }
    u64 x = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) x ^= st[k];
    out[i] = x;
}

// The same permutation with ONE state word per lane, for the handful of permutations the Fiat-Shamir transcript asks for
// one at a time (transcript.cpp:23,46 -- ~800 per zkEVM proof, each waited for by the host): a lane that walks the whole
// permutation alone retires ~21 500 dependent instructions (64 us measured); here the 12 S-boxes of a full round run in 12
// lanes and every lane builds its own row of the MDS product from the other lanes' words (v_readlane broadcasts against the
// lane's row of the circulant), ~6x fewer instructions on the critical path.  One wave per permutation; lanes 12..63 mirror
// lanes 0..11 (i = lane % 12) so that no lane diverges outside the partial rounds.
__constant__ u32 MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
// one permutation, the state distributed over the lanes (word i = lane % 12 in every lane): in = that lane's word, any 64-bit value
__device__ __forceinline__ u64 perm_coop_lanes(u64 in, const u32 (&m)[12], u32 i)
{
    u64 x = gl::add(in, POSEIDON_C[i]);                    // any + canon
#pragma unroll 1
    for (int r = 0; r < 30; r++) {
        if (r < 4 || r >= 26 || i == 0) x = pow7(x);
        u64 L = 0, H = 0;
        if (r < 29) { const u64 c = POSEIDON_C[12 * (r + 1) + i]; L = (u32)c; H = c >> 32; }
        const u32 xl = (u32)x, xh = (u32)(x >> 32);
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const u32 bl = (u32)__builtin_amdgcn_readlane((int)xl, j), bh = (u32)__builtin_amdgcn_readlane((int)xh, j);
            L += (u64)bl * m[j];
            H += (u64)bh * m[j];
        }
        // L + H * 2^32 (mod p), as in mds()
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
        x = gl::pack(rlo, rhi);
    }
    return x;
}
__global__ __launch_bounds__(64) void perm_coop_kernel(u64* out, const u64* in, u64 count)
{
    const u32 lane = threadIdx.x, i = lane % 12;
    u32 m[12];                                               // row i of M = circ + diag(8, 0, ...)
#pragma unroll
    for (int j = 0; j < 12; j++) m[j] = MDS_CIRC[(j + 12 - i) % 12] + ((i == 0 && j == 0) ? 8u : 0u);
    for (u64 p = blockIdx.x; p < count; p += gridDim.x) {
        const u64 x = perm_coop_lanes(in[12 * p + i], m, i);
        if (lane < 12) out[12 * p + i] = gl::canon(x);
    }
}

// The transcript absorbing a long message (Transcript::put of the 3 x 1 973 evaluation words of a zkEVM proof, transcript.cpp:12-29): nblocks
// dependent permutations, block b = perm(elems[8b .. 8b + 8) | capacity), capacity = words 0..3 of the previous output.  One wave walks the
// whole chain (739 permutations: ~4 ms) instead of one launch, two copies and a host wait per permutation.
__global__ __launch_bounds__(64) void perm_chain_kernel(u64* out, const u64* state, const u64* elems, u64 nblocks)
{
    const u32 lane = threadIdx.x, i = lane % 12;
    u32 m[12];
#pragma unroll
    for (int j = 0; j < 12; j++) m[j] = MDS_CIRC[(j + 12 - i) % 12] + ((i == 0 && j == 0) ? 8u : 0u);
    u64 cap = i >= 8 ? state[i - 8] : 0, x = 0;
    const u32 from = i >= 8 ? i - 8 : i;                     // lanes 8..11 take words 0..3 of the output as the next capacity
    for (u64 b = 0; b < nblocks; b++) {
        x = perm_coop_lanes(i < 8 ? elems[8 * b + i] : cap, m, i);
        const u32 lo = (u32)__shfl((int)(u32)x, (int)from), hi = (u32)__shfl((int)(u32)(x >> 32), (int)from);
        cap = gl::pack(lo, hi);
    }
    if (lane < 12) out[i] = gl::canon(x);
}

// Leaf level: out[4*row..] = linear_hash(src[row][0..w)).
// w <= 4: copy + zero-pad, no permutation.
__global__ __launch_bounds__(256) void linear_hash_kernel(u64* out, const u64* src, u64 w, u64 h)
{
    u64 row = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= h) return;
    const u64* in = src + row * w;
    u64 st[12];
    if (w <= 4) {
        for (u64 k = 0; k < 4; k++) out[4 * row + k] = k < w ? gl::canon(in[k]) : 0;
        return;
    }
    st[8] = st[9] = st[10] = st[11] = 0;
    u64 rem = w;
    while (rem) {
        if (rem >= 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) st[k] = in[k];
            in += 8; rem -= 8;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) st[k] = (u64)k < rem ? in[k] : 0;
            rem = 0;
        }
        permute(st);
        st[8] = st[0]; st[9] = st[1]; st[10] = st[2]; st[11] = st[3];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) out[4 * row + k] = st[k];
}

// The same sponge continued over a band of columns: state[4*row..] carries the capacity between calls (it IS the leaf once the
// last band is absorbed).  Band = columns [first, first + wc) of a row-major matrix of row stride ld; first is a multiple of
// 8 (a sponge block never straddles two bands) and only the last band may have wc % 8 != 0 (zero-padded like the tail of
// linear_hash).  Absorbing the bands of a row in order gives exactly linear_hash(row): capacity 0^4 before the first block,
// out[0..3] of the previous block afterwards.
__global__ __launch_bounds__(256) void linear_hash_absorb_kernel(u64* state, const u64* src, u64 ld, u64 first, u64 wc, u64 h, int is_first)
{
    u64 row = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= h) return;
    const u64* in = src + row * ld + first;
    u64 st[12];
    if (is_first) { st[8] = st[9] = st[10] = st[11] = 0; }
    else {
#pragma unroll
        for (int k = 0; k < 4; k++) st[8 + k] = state[4 * row + k];
    }
    u64 rem = wc;
    while (rem) {
        if (rem >= 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) st[k] = in[k];
            in += 8; rem -= 8;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) st[k] = (u64)k < rem ? in[k] : 0;
            rem = 0;
        }
        permute(st);
        st[8] = st[0]; st[9] = st[1]; st[10] = st[2]; st[11] = st[3];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) state[4 * row + k] = st[k];
}

// The leaf sponge on a COLUMN-MAJOR section (colmajor.hip): lane = dense coset-major index d = r * nq + q of an extended row, i.e. storage
// row r * cos + q of every column, so a wave reads 512 contiguous bytes of one column per absorbed word (the row-major sponge reads 64-byte
// pieces of rows up to 6 KB apart: 2.1x the algorithmic traffic).  The leaf belongs to NATURAL row k = (q << ext) | r: state[4 k ..].
// Bands as in linear_hash_absorb_kernel: src = the section's column `first`, wc columns, first % 8 == 0, wc % 8 == 0 unless last.
__global__ __launch_bounds__(256) void linear_hash_absorb_cm_kernel(u64* state, const u64* src, u64 cstride, u64 cos, u64 wc, u64 h, u32 ext, u32 lognq, int is_first)
{
    const u64 d = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= h) return;
    const u64 r = d >> lognq, q = d & ((1ull << lognq) - 1);
    const u64 k = (q << ext) | r;
    const u64* in = src + r * cos + q;
    u64 st[12];
    if (is_first) { st[8] = st[9] = st[10] = st[11] = 0; }
    else {
#pragma unroll
        for (int j = 0; j < 4; j++) st[8 + j] = state[4 * k + j];
    }
    u64 rem = wc;
    while (rem) {
        if (rem >= 8) {
#pragma unroll
            for (int j = 0; j < 8; j++) st[j] = in[j * cstride];
            in += 8 * cstride; rem -= 8;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) st[j] = (u64)j < rem ? in[j * cstride] : 0;
            rem = 0;
        }
        permute(st);
        st[8] = st[0]; st[9] = st[1]; st[10] = st[2]; st[11] = st[3];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) state[4 * k + j] = st[j];
}
// w <= 4: the leaf is the row itself, zero-padded (no hash)
__global__ __launch_bounds__(256) void leaf_copy_cm_kernel(u64* out, const u64* src, u64 cstride, u64 cos, u64 w, u64 h, u32 ext, u32 lognq)
{
    const u64 d = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= h) return;
    const u64 r = d >> lognq, q = d & ((1ull << lognq) - 1);
    const u64 k = (q << ext) | r;
    for (u64 j = 0; j < 4; j++) out[4 * k + j] = j < w ? gl::canon(src[j * cstride + r * cos + q]) : 0;
}

// One tree level: nxt[j] = perm(cur[2j] | cur[2j+1] | 0^4)[0..3].
__global__ __launch_bounds__(256) void merkle_level_kernel(u64* nxt, const u64* cur, u64 count)
{
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    u64 st[12];
#pragma unroll
    for (int k = 0; k < 8; k++) st[k] = cur[8 * j + k];
    st[8] = st[9] = st[10] = st[11] = 0;
    permute(st);
#pragma unroll
    for (int k = 0; k < 4; k++) nxt[4 * j + k] = st[k];
}

int poseidon_init(Ctx*) { return 0; }

int dev_perm(Ctx* ctx, u64* out, const u64* in, u64 count)
{
    if (!count) return 0;
    if (count <= 256) {                                       // too few to fill lanes: one wave per permutation, 12 lanes per state
        hipLaunchKernelGGL(perm_coop_kernel, dim3((unsigned)count), dim3(64), 0, ctx->stream, out, in, count);
        XLP_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(perm_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, out, in, count);
    XLP_HIP(hipGetLastError());
    return 0;
}

int dev_perm_resident(Ctx* ctx, u64* out, u64 count, u64 iters)
{
    if (!count) return 0;
    hipLaunchKernelGGL(perm_resident_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, out, count, iters);
    XLP_HIP(hipGetLastError());
    return 0;
}
int dev_perm_chain(Ctx* ctx, u64* out, const u64* state, const u64* elems, u64 nblocks)
{
    if (!nblocks) return 0;
    hipLaunchKernelGGL(perm_chain_kernel, dim3(1), dim3(64), 0, ctx->stream, out, state, elems, nblocks);
    XLP_HIP(hipGetLastError());
    return 0;
}

int dev_linear_hash(Ctx* ctx, u64* out, const u64* src, u64 w, u64 h)
{
    if (!h) return 0;
    if (w == 0) { XLP_HIP(hipMemsetAsync(out, 0, h * 4 * sizeof(u64), ctx->stream)); return 0; }
    hipLaunchKernelGGL(linear_hash_kernel, dim3((unsigned)((h + 255) / 256)), dim3(256), 0, ctx->stream, out, src, w, h);
    XLP_HIP(hipGetLastError());
    return 0;
}

int dev_linear_hash_absorb(Ctx* ctx, u64* state, const u64* src, u64 ld, u64 first, u64 ncols, u64 h, int is_first, int is_last)
{
    if (!h || !ncols) return 0;
    if (first % 8) return xlp_fail("linear hash (band): the band must start at a multiple of 8 columns");
    if (!is_last && ncols % 8) return xlp_fail("linear hash (band): only the last band may end inside a sponge block");
    if (first + ncols > ld) return xlp_fail("linear hash (band): band exceeds the row stride");
    if (is_first && is_last && ncols <= 4) return xlp_fail("linear hash (band): rows of <= 4 columns are not hashed (use xlp_dev_linear_hash)");
    hipLaunchKernelGGL(linear_hash_absorb_kernel, dim3((unsigned)((h + 255) / 256)), dim3(256), 0, ctx->stream, state, src, ld, first, ncols, h, is_first);
    XLP_HIP(hipGetLastError());
    return 0;
}

// PoseidonGoldilocks::merkletree_avx(tree, input, ncols, nrows): nodes =
// level0 | level1 | ... | root, (2h-1)*4 elements (merkleTreeGL.hpp:61-68).
static int cm_geometry(const char* what, u64 h, u32 ext, u64& cos, u32& lognq)
{
    if (h == 0 || (h & (h - 1))) return xlp_fail((std::string(what) + ": number of rows is not a power of two").c_str());
    if (ext > 8 || (h >> ext) == 0) return xlp_fail((std::string(what) + ": more cosets than rows").c_str());
    const u64 nq = h >> ext;
    lognq = 0;
    while ((1ull << lognq) < nq) lognq++;
    if (cos == 0) cos = nq;
    if (cos < nq) return xlp_fail((std::string(what) + ": coset stride smaller than the rows of a coset").c_str());
    return 0;
}
int dev_linear_hash_absorb_cm(Ctx* ctx, u64* state, const u64* src, u64 cstride, u64 cos, u64 first, u64 ncols, u64 h, u32 ext, int is_first, int is_last)
{
    if (!h || !ncols) return 0;
    u32 lognq;
    if (cm_geometry("linear hash (column-major)", h, ext, cos, lognq)) return -1;
    if (first % 8) return xlp_fail("linear hash (column-major): a band must start on a multiple of 8 columns");
    if (!is_last && ncols % 8) return xlp_fail("linear hash (column-major): only the last band may hold a partial block");
    hipLaunchKernelGGL(linear_hash_absorb_cm_kernel, dim3((unsigned)((h + 255) / 256)), dim3(256), 0, ctx->stream, state, src + first * cstride, cstride, cos, ncols, h, ext, lognq, is_first);
    XLP_HIP(hipGetLastError());
    return 0;
}
int dev_merkle_cm(Ctx* ctx, u64* nodes, const u64* src, u64 cstride, u64 cos, u64 w, u64 h, u32 ext)
{
    if (!h) return 0;
    u32 lognq;
    if (cm_geometry("merkle (column-major)", h, ext, cos, lognq)) return -1;
    if (w <= 4) {
        hipLaunchKernelGGL(leaf_copy_cm_kernel, dim3((unsigned)((h + 255) / 256)), dim3(256), 0, ctx->stream, nodes, src, cstride, cos, w, h, ext, lognq);
        XLP_HIP(hipGetLastError());
    } else if (dev_linear_hash_absorb_cm(ctx, nodes, src, cstride, cos, 0, w, h, ext, 1, 1)) return -1;
    return dev_merkle_levels(ctx, nodes, h);
}

int dev_merkle(Ctx* ctx, u64* nodes, const u64* src, u64 w, u64 h)
{
    if (!h) return 0;
    if (h & (h - 1)) return xlp_fail("merkletree: number of rows is not a power of two");
    if (dev_linear_hash(ctx, nodes, src, w, h)) return -1;
    return dev_merkle_levels(ctx, nodes, h);
}

// the levels above the leaves: nodes[0 .. 4h) already holds level 0
int dev_merkle_levels(Ctx* ctx, u64* nodes, u64 h)
{
    if (!h) return 0;
    if (h & (h - 1)) return xlp_fail("merkletree: number of rows is not a power of two");
    u64 off = 0, cnt = h;
    while (cnt > 1) {
        u64 half = cnt / 2;
        hipLaunchKernelGGL(merkle_level_kernel, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, ctx->stream, nodes + off + 4 * cnt,
                           nodes + off, half);
        off += 4 * cnt;
        cnt = half;
    }
    XLP_HIP(hipGetLastError());
    return 0;
}

// MerkleTreeGL::getGroupProof (merkleTreeGL.cpp:12-35) for many indices at once: out[q] = row idx_q (w words) followed by
// the sibling nodes bottom-up (4 words per level); one block per query, so a proof with hundreds of queries costs one
// kernel and one copy instead of (1 + log2 h) small copies per query.
__global__ __launch_bounds__(128) void merkle_proofs_kernel(u64* out, const u64* src, const u64* nodes, u64 w, u64 h, const u64* idxs, u32 levels)
{
    const u64 idx = idxs[blockIdx.x];
    u64* o = out + (u64)blockIdx.x * (w + 4ull * levels);
    for (u64 i = threadIdx.x; i < w; i += blockDim.x) o[i] = src[idx * w + i];
    if (threadIdx.x < 4 * levels) {
        const u32 l = threadIdx.x >> 2, c = threadIdx.x & 3;
        u64 off = 0, cnt = h;
        for (u32 k = 0; k < l; k++) { off += cnt * 4; cnt >>= 1; }
        o[w + 4 * l + c] = nodes[off + (((idx >> l) ^ 1) * 4) + c];
    }
}

// the same on a column-major section: the opened row is gathered from the w columns (storage row r * cos + q of natural row (q << ext) | r)
__global__ __launch_bounds__(128) void merkle_proofs_cm_kernel(u64* out, const u64* src, u64 cstride, u64 cos, const u64* nodes, u64 w, u64 h, u32 ext, const u64* idxs, u32 levels)
{
    const u64 idx = idxs[blockIdx.x];
    const u64 s = (idx & ((1ull << ext) - 1)) * cos + (idx >> ext);
    u64* o = out + (u64)blockIdx.x * (w + 4ull * levels);
    for (u64 i = threadIdx.x; i < w; i += blockDim.x) o[i] = src[i * cstride + s];
    if (threadIdx.x < 4 * levels) {
        const u32 l = threadIdx.x >> 2, c = threadIdx.x & 3;
        u64 off = 0, cnt = h;
        for (u32 k = 0; k < l; k++) { off += cnt * 4; cnt >>= 1; }
        o[w + 4 * l + c] = nodes[off + (((idx >> l) ^ 1) * 4) + c];
    }
}
int dev_merkle_proofs_cm(Ctx* ctx, u64* out, const u64* src, u64 cstride, u64 cos, const u64* nodes, u64 w, u64 h, u32 ext, const u64* d_idxs, u64 nq)
{
    if (!nq) return 0;
    u32 lognq;
    if (cm_geometry("merkle proofs (column-major)", h, ext, cos, lognq)) return -1;
    u32 levels = 0;
    while ((1ull << levels) < h) levels++;
    if (levels > 32) return xlp_fail("merkle proofs: tree too tall");
    hipLaunchKernelGGL(merkle_proofs_cm_kernel, dim3((unsigned)nq), dim3(128), 0, ctx->stream, out, src, cstride, cos, nodes, w, h, ext, d_idxs, levels);
    XLP_HIP(hipGetLastError());
    return 0;
}

int dev_merkle_proofs(Ctx* ctx, u64* out, const u64* src, const u64* nodes, u64 w, u64 h, const u64* d_idxs, u64 nq)
{
    if (!nq) return 0;
    if (h == 0 || (h & (h - 1))) return xlp_fail("merkle proofs: number of rows is not a power of two");
    u32 levels = 0;
    while ((1ull << levels) < h) levels++;
    if (levels > 32) return xlp_fail("merkle proofs: tree too tall");
    hipLaunchKernelGGL(merkle_proofs_kernel, dim3((unsigned)nq), dim3(128), 0, ctx->stream, out, src, nodes, w, h, d_idxs, levels);
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
