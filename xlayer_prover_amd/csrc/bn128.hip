// BN128 side of StarkRecursiveF for gfx950: Poseidon over the BN254 scalar field (circomlib's optimized
// form), the 16-ary MerkleTreeBN128 and the permutation TranscriptBN128 sits on.
//
// Replaces, on the GPU:
//   Poseidon_opt::hash                     src/poseidon_opt/poseidon_opt.cpp:3-100   (constants: constants_opt.hpp, as data)
//   MerkleTreeBN128::linearHash/merkelize  src/starkpil/merkleTree/merkleTreeBN128.cpp:89-186
//   RawFr (ffiasm fr.asm) mul/add/toMontgomery as those two use them
//
// Design.  An Fr element is 8 x 32-bit limbs in Montgomery form (R = 2^256, the reference's in-memory representation, so
// node buffers are byte-identical).  One hash per lane; a workgroup is ONE wave.  The state (up to 17 elements = 136 words per
// lane) lives in LDS as st[element][limb][lane] (conflict-free, 34 KiB per wave -> four waves per CU, one per SIMD); on CDNA a
// single wave keeps its SIMD's integer pipe busy as long as it does not wait for memory, and the round constants and matrices
// are wave-uniform (scalar loads).  The MDS mix walks the state once (j outer) and keeps all t accumulators in registers (one
// wave per SIMD has the whole 512-entry register file), so every matrix element and every state word is read once per mix.
// The state size t (2..17) is a run-time parameter: a leaf's last chunk hashes pending + 1 elements (merkleTreeBN128.cpp:131-137).
// Integer-VALU bound like the Goldilocks sponge: one Montgomery product = 128 v_mad_u64_u32 + carries.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <dlfcn.h>
#include <string>
#include <vector>
#include "xlp_internal.hpp"
#include "mont256.hpp"

namespace xlp {
namespace bn {

struct Fr { u32 v[8]; };

// r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
__device__ __constant__ u32 R_MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
// R^2 mod r (toMontgomery = mul by R^2)
__device__ __constant__ u32 R2_MOD[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};

__device__ __forceinline__ Fr zero() { Fr z; for (int i = 0; i < 8; i++) z.v[i] = 0; return z; }

// a - r if a >= r  (a < 2r, possibly with a 9th limb `hi`)
__device__ __forceinline__ Fr cond_sub(const u32 (&a)[8], u32 hi)
{
    u32 d[8];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 s = (u64)a[i] - R_MOD[i] - borrow;
        d[i] = (u32)s;
        borrow = (u32)(s >> 63);
    }
    const bool ge = hi != 0 || borrow == 0;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = ge ? d[i] : a[i];
    return r;
}

__device__ __forceinline__ Fr add(const Fr& a, const Fr& b)
{
    u32 s[8];
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 t = (u64)a.v[i] + b.v[i] + c;
        s[i] = (u32)t;
        c = (u32)(t >> 32);
    }
    return cond_sub(s, c);      // r < 2^254: c is always 0, kept for the general shape
}

// Montgomery product a * b / R mod r (mont256.hpp: product scanning, 2 VALU per multiply-accumulate).  b < r; a < 2^256
// (toMontgomery of a raw 4-limb word relies on it).
struct FrParams {
    static constexpr u32 N[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr u32 N0INV = 0xefffffffu;
};
__device__ __forceinline__ Fr mul(const Fr& a, const Fr& b)          // Poseidon kernels: one wave per SIMD -> split accumulators
{
    Fr r;
    mont::mul<FrParams, true>(r.v, a.v, b.v);
    return r;
}
__device__ __forceinline__ Fr mul1(const Fr& a, const Fr& b)         // FFT kernels: occupancy hides the latency
{
    Fr r;
    mont::mul<FrParams, false>(r.v, a.v, b.v);
    return r;
}

__device__ __forceinline__ Fr exp5(const Fr& x)
{
    const Fr x2 = mul(x, x);
    const Fr x4 = mul(x2, x2);
    return mul(x4, x);
}

__device__ __forceinline__ Fr to_montgomery(const Fr& raw)
{
    Fr r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = R2_MOD[i];
    return mul(raw, r2);
}

// ----------------------------------------------------------------- constants (device copy of poseidon_bn128.bin)
struct Tables {
    const Fr* C[16];
    const Fr* S[16];
    const Fr* M[16];
    const Fr* P[16];
    u32 nP[16];
};

constexpr int WAVE = 64;
constexpr int MAXT = 17;
// kernels are built for two state-size classes: MT = 8 (t <= 8: 40-64 accumulator registers, 16 KiB of LDS per wave -> two waves per
// SIMD; the leaves of the recursivef trees, t = 5 and t = 8) and MT = 17 (everything)

// state in LDS: st[(j * 8 + l) * WAVE + lane]
__device__ __forceinline__ Fr ld(const u32* st, int j)
{
    Fr e;
#pragma unroll
    for (int l = 0; l < 8; l++) e.v[l] = st[(j * 8 + l) * WAVE];
    return e;
}
__device__ __forceinline__ void sst(u32* st, int j, const Fr& e)
{
#pragma unroll
    for (int l = 0; l < 8; l++) st[(j * 8 + l) * WAVE] = e.v[l];
}
__device__ __forceinline__ Fr ldc(const Fr* __restrict__ tab, u32 idx)   // wave-uniform table element
{
    Fr e;
#pragma unroll
    for (int l = 0; l < 8; l++) e.v[l] = tab[idx].v[l];
    return e;
}

// new[i] = sum_j mat[j * t + i] * st[j]   (Poseidon_opt::mix, poseidon_opt.cpp:89-100), every output accumulated UNREDUCED (16 limbs:
// a sum of t <= 17 products of values below r stays below 2^511.3) and reduced ONCE: t x 64 multiply-accumulates + 64 per output
// instead of t x 128, and no modular additions.  Round 2: recursivef tree 2^23 x 12 0.195 -> 0.135 s, t = 17 leaves 0.142 -> 0.093 s.
__device__ __forceinline__ void ldk(u32 (&k)[8], const Fr* __restrict__ tab, u32 idx)
{
#pragma unroll
    for (int l = 0; l < 8; l++) k[l] = tab[idx].v[l];
}
template <int SUBS>
__device__ __forceinline__ Fr redc(const u32 (&acc)[16])
{
    Fr r;
    mont::redc512<FrParams, SUBS>(r.v, acc);
    return r;
}
// The outputs are built one after the other (i outer: 17 x 16 accumulator registers would not fit), each from the whole state re-read
// from LDS (8 ds_read per 64 multiply-accumulates) into one unreduced accumulator; the new state goes to registers first.  (Keeping
// all accumulators of a t <= 8 state in registers, j outer, measured the same.)
template <int MT>
__device__ __forceinline__ void mix_lazy_rows(u32* st, const int t, const Fr* __restrict__ mat)
{
    Fr out[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) {
        if (i < t) {
            u32 acc[16];
#pragma unroll
            for (int l = 0; l < 16; l++) acc[l] = 0;
#pragma unroll 1
            for (int j = 0; j < t; j++) {
                const Fr s = ld(st, j);
                u32 k[8];
                ldk(k, mat, (u32)(j * t + i));
                mont::mac512(acc, k, s.v);
            }
            out[i] = t <= 5 ? redc<1>(acc) : t <= 10 ? redc<2>(acc) : redc<4>(acc);
        }
    }
#pragma unroll
    for (int i = 0; i < MT; i++)
        if (i < t) sst(st, i, out[i]);
}

// Poseidon_opt::hash(vector<FrElement>& state) (poseidon_opt.cpp:9-66) on the lane's LDS state, t = 2..17
template <int MT>
__device__ __forceinline__ void permute(u32* st, const int t, const Tables& K)
{
    const int k = t - 2;
    const Fr* __restrict__ c = K.C[k];
    const Fr* __restrict__ s = K.S[k];
    const int nP = (int)K.nP[k];
#pragma unroll 1
    for (int i = 0; i < t; i++) sst(st, i, add(ld(st, i), ldc(c, (u32)i)));
#pragma unroll 1
    for (int fr = 0; fr < 8; fr++) {
        if (fr == 4) {
#pragma unroll 1
            for (int r = 0; r < nP; r++) {
                const Fr x = add(exp5(ld(st, 0)), ldc(c, (u32)(5 * t + r)));
                const u32 base = (u32)((2 * t - 1) * r);
                // new state[0] = sum_j S[j] * state[j]: accumulated unreduced (t <= 17 products of values below r: < 2^511.3), reduced once
                u32 a0[16];
#pragma unroll
                for (int l = 0; l < 16; l++) a0[l] = 0;
                { u32 k[8]; ldk(k, s, base); mont::mac512(a0, k, x.v); }
#pragma unroll 1
                for (int j = 1; j < t; j++) {
                    Fr sj = ld(st, j);
                    { u32 k[8]; ldk(k, s, base + (u32)j); mont::mac512(a0, k, sj.v); }
                    sj = add(sj, mul(x, ldc(s, base + (u32)(t + j - 1))));
                    sst(st, j, sj);
                }
                sst(st, 0, t <= 5 ? redc<1>(a0) : t <= 10 ? redc<2>(a0) : redc<4>(a0));   // t r / 2^256 + 1 < 2, 3, 5
            }
        }
        const u32 cidx = fr < 4 ? (u32)((fr + 1) * t) : (u32)(5 * t + nP + (fr - 4) * t);
#pragma unroll 1
        for (int i = 0; i < t; i++) {
            Fr x = exp5(ld(st, i));
            if (fr != 7) x = add(x, ldc(c, cidx + (u32)i));
            sst(st, i, x);
        }
        mix_lazy_rows<MT>(st, t, fr == 3 ? K.P[k] : K.M[k]);
    }
}

// ----------------------------------------------------------------- kernels
// leaves: MerkleTreeBN128::linearHash (merkleTreeBN128.cpp:89-157), one row per lane
// (MT = 5: rows of up to 12 columns -- recursivef's cm1 -- hash with t = 5; 10 KB of LDS per wavefront: 16 of them per CU instead of 10 with MT = 8)
template <int MT>
__global__ __launch_bounds__(WAVE, MT <= 5 ? 3 : MT <= 8 ? 2 : 1) void bn128_leaf_kernel(Fr* __restrict__ nodes, const u64* __restrict__ src, const u64 w, const u64 h, const Tables K)
{
    __shared__ u32 lds[MT * 8 * WAVE];
    u32* st = lds + threadIdx.x;
    const u64 gid = (u64)blockIdx.x * WAVE + threadIdx.x;
    const u64 row = gid < h ? gid : h - 1;            // idle lanes repeat the last row and do not store
    const u64* __restrict__ p = src + row * w;
    if (w <= 4) {
        Fr raw = zero();
        for (u64 k = 0; k < w; k++) { raw.v[2 * k] = (u32)p[k]; raw.v[2 * k + 1] = (u32)(p[k] >> 32); }
        const Fr m = to_montgomery(raw);
        if (gid < h) nodes[gid] = m;
        return;
    }
    const u64 nfr = (w + 2) / 3;
    Fr node = zero();
    u64 done = 0;
    while (done < nfr) {
        const u64 take = nfr - done < 16 ? nfr - done : 16;
        sst(st, 0, node);
        for (u64 e = 0; e < take; e++) {
            Fr raw = zero();
            const u64 c0 = (done + e) * 3;
            for (u64 q = 0; q < 3 && c0 + q < w; q++) {
                const u64 x = p[c0 + q];
                raw.v[2 * q] = (u32)x;
                raw.v[2 * q + 1] = (u32)(x >> 32);
            }
            sst(st, 1 + (int)e, to_montgomery(raw));
        }
        permute<MT>(st, (int)take + 1, K);
        node = ld(st, 0);
        done += take;
    }
    if (gid < h) nodes[gid] = node;
}

// one level of MerkleTreeBN128::merkelize (merkleTreeBN128.cpp:167-185): parent i = hash(0, children[16 i .. 16 i + num))
__global__ __launch_bounds__(WAVE) void bn128_level_kernel(Fr* __restrict__ next, const Fr* __restrict__ cur, const u64 batches, const u32 num, const Tables K)
{
    __shared__ u32 lds[MAXT * 8 * WAVE];
    u32* st = lds + threadIdx.x;
    const u64 gid = (u64)blockIdx.x * WAVE + threadIdx.x;
    const u64 b = gid < batches ? gid : batches - 1;
    sst(st, 0, zero());
    for (u32 e = 0; e < 16; e++) {
        Fr x = zero();
        if (e < num) x = cur[b * 16 + e];
        sst(st, 1 + (int)e, x);
    }
    permute<MAXT>(st, 17, K);
    if (gid < batches) next[gid] = ld(st, 0);
}

// count independent permutations of t-element states (Montgomery limbs in and out): the transcript's and the tests' entry
template <int MT>
__global__ __launch_bounds__(WAVE, MT <= 8 ? 2 : 1) void bn128_perm_kernel(Fr* __restrict__ states, const u32 t, const u64 count, const Tables K)
{
    __shared__ u32 lds[MT * 8 * WAVE];
    u32* st = lds + threadIdx.x;
    const u64 gid = (u64)blockIdx.x * WAVE + threadIdx.x;
    const u64 b = gid < count ? gid : count - 1;
    for (u32 e = 0; e < t; e++) sst(st, (int)e, states[b * t + e]);
    permute<MT>(st, (int)t, K);
    if (gid < count)
        for (u32 e = 0; e < t; e++) states[b * t + e] = ld(st, (int)e);
}

// ----------------------------------------------------------------- FFT over Fr (src/ffiasm/fft.c.hpp:181-221, the fflonk prover's transform)
__device__ __forceinline__ Fr sub(const Fr& a, const Fr& b)
{
    u32 d[8];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 s = (u64)a.v[i] - b.v[i] - borrow;
        d[i] = (u32)s;
        borrow = (u32)(s >> 63);
    }
    // borrow: add r back
    u32 c = 0;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 t = (u64)d[i] + (borrow ? R_MOD[i] : 0u) + c;
        r.v[i] = (u32)t;
        c = (u32)(t >> 32);
    }
    return r;
}
__device__ __forceinline__ Fr gld(const Fr* p)
{
    const uint4* q = (const uint4*)p;
    const uint4 a = q[0], b = q[1];
    Fr e;
    e.v[0] = a.x; e.v[1] = a.y; e.v[2] = a.z; e.v[3] = a.w; e.v[4] = b.x; e.v[5] = b.y; e.v[6] = b.z; e.v[7] = b.w;
    return e;
}
__device__ __forceinline__ void gst(Fr* p, const Fr& e)
{
    uint4* q = (uint4*)p;
    q[0] = make_uint4(e.v[0], e.v[1], e.v[2], e.v[3]);
    q[1] = make_uint4(e.v[4], e.v[5], e.v[6], e.v[7]);
}

// tw[j] = w^j, j < count, from wpow[k] = w^(2^k)
__global__ void fr_twiddle_kernel(Fr* __restrict__ tw, const u64 count, const Fr* __restrict__ wpow, const Fr one)
{
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fr acc = one;
    for (u32 k = 0; (j >> k) != 0; k++)
        if ((j >> k) & 1) acc = mul1(acc, gld(&wpow[k]));
    gst(&tw[j], acc);
}

__device__ __forceinline__ u64 brev_bits(u64 x, u32 bits) { return bits ? __brevll(x) >> (64 - bits) : 0; }

// FFT::reversePermutation (fft.c.hpp:166-178)
__global__ void fr_bitrev_kernel(Fr* __restrict__ a, const u32 logn)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> logn) return;
    const u64 r = brev_bits(i, logn);
    if (i > r) { const Fr x = gld(&a[i]), y = gld(&a[r]); gst(&a[i], y); gst(&a[r], x); }
}

// stages 1..LB of FFT::fft on one block of 2^LB consecutive elements held in LDS (the butterflies of those stages stay inside the block)
constexpr int FFT_LB = 8;
__global__ __launch_bounds__(1 << (FFT_LB - 1)) void fr_fft_low_kernel(Fr* __restrict__ a, const Fr* __restrict__ tw, const u32 tl, const u32 nst)
{
    __shared__ u32 sm[8 << FFT_LB];                 // [limb][element]
    const u32 tid = threadIdx.x, B = 1u << nst;
    Fr* blk = a + ((u64)blockIdx.x << nst);
    for (u32 e = tid; e < B; e += blockDim.x) {
        const Fr x = gld(&blk[e]);
#pragma unroll
        for (int l = 0; l < 8; l++) sm[l * (1 << FFT_LB) + e] = x.v[l];
    }
    __syncthreads();
    for (u32 s = 1; s <= nst; s++) {
        const u32 mdiv2 = 1u << (s - 1);
        for (u32 i = tid; i < (B >> 1); i += blockDim.x) {
            const u32 k = (i / mdiv2) << s, j = i % mdiv2;
            Fr u, v;
#pragma unroll
            for (int l = 0; l < 8; l++) { u.v[l] = sm[l * (1 << FFT_LB) + k + j]; v.v[l] = sm[l * (1 << FFT_LB) + k + j + mdiv2]; }
            const Fr t = mul1(gld(&tw[(u64)j << (tl - s)]), v);          // root(s, j) in the table of the 2^tl-th root
            const Fr x = add(u, t), y = sub(u, t);
#pragma unroll
            for (int l = 0; l < 8; l++) { sm[l * (1 << FFT_LB) + k + j] = x.v[l]; sm[l * (1 << FFT_LB) + k + j + mdiv2] = y.v[l]; }
        }
        __syncthreads();
    }
    for (u32 e = tid; e < B; e += blockDim.x) {
        Fr x;
#pragma unroll
        for (int l = 0; l < 8; l++) x.v[l] = sm[l * (1 << FFT_LB) + e];
        gst(&blk[e], x);
    }
}

// stages s0 + 1 .. s0 + nst (nst <= 8) of FFT::fft in ONE pass through HBM: index = hi * 2^(s0 + nst) + m * 2^s0 + lo; a workgroup holds, for one hi,
// all 2^nst values of m for T consecutive lo (T * 2^nst = 2^FFT_MB elements in LDS, rows of T x 32 bytes in HBM) -- the butterflies of those stages pair
// values of m only.  (Until round 4 these stages went two at a time through HBM: 8 passes over 2^24 elements instead of 2; 4.35 -> 3.62 ms, the rest is the
// 24 x 2^23 Montgomery products themselves.)
constexpr int FFT_MB = 10;
__global__ __launch_bounds__(256) void fr_fft_mid_kernel(Fr* __restrict__ a, const Fr* __restrict__ tw, const u32 s0, const u32 nst, const u32 lt, const u32 tl)
{
    __shared__ u32 sm[8 << FFT_MB];                 // [limb][m * T + l]
    const u32 tid = threadIdx.x, T = 1u << lt, E = 1u << (nst + lt);
    const u64 tiles = 1ull << (s0 - lt);            // tiles of T consecutive lo per hi
    const u64 hi = blockIdx.x / tiles, lo0 = (blockIdx.x % tiles) << lt;
    Fr* base = a + (hi << (s0 + nst)) + lo0;
    for (u32 e = tid; e < E; e += 256) {
        const Fr x = gld(&base[((u64)(e >> lt) << s0) + (e & (T - 1))]);
#pragma unroll
        for (int l = 0; l < 8; l++) sm[l * (1 << FFT_MB) + e] = x.v[l];
    }
    __syncthreads();
    for (u32 t = 1; t <= nst; t++) {
        const u32 half = 1u << (t - 1);
        for (u32 i = tid; i < (E >> 1); i += 256) {
            const u32 im = i >> lt, l = i & (T - 1);
            const u32 mlow = im & (half - 1), m = ((im >> (t - 1)) << t) | mlow;
            const u32 e0 = (m << lt) | l, e1 = e0 + (half << lt);
            Fr u, v;
#pragma unroll
            for (int k = 0; k < 8; k++) { u.v[k] = sm[k * (1 << FFT_MB) + e0]; v.v[k] = sm[k * (1 << FFT_MB) + e1]; }
            const u64 j = ((u64)mlow << s0) + lo0 + l;                                       // position inside the half-group of stage s0 + t
            const Fr w = mul1(gld(&tw[j << (tl - s0 - t)]), v);
            const Fr x = add(u, w), y = sub(u, w);
#pragma unroll
            for (int k = 0; k < 8; k++) { sm[k * (1 << FFT_MB) + e0] = x.v[k]; sm[k * (1 << FFT_MB) + e1] = y.v[k]; }
        }
        __syncthreads();
    }
    for (u32 e = tid; e < E; e += 256) {
        Fr x;
#pragma unroll
        for (int l = 0; l < 8; l++) x.v[l] = sm[l * (1 << FFT_MB) + e];
        gst(&base[((u64)(e >> lt) << s0) + (e & (T - 1))], x);
    }
}

// the tail of FFT::ifft (fft.c.hpp:205-221): a[i] <-> a[n - i], everything times 1 / n
__global__ void fr_ifft_finish_kernel(Fr* __restrict__ a, const u32 logn, const Fr ninv)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 n = 1ull << logn, half = n >> 1;
    if (i > half) return;
    if (i == 0 || i == half) { gst(&a[i], mul1(gld(&a[i]), ninv)); return; }
    const Fr x = gld(&a[i]), y = gld(&a[n - i]);
    gst(&a[i], mul1(y, ninv));
    gst(&a[n - i], mul1(x, ninv));
}

// MerkleTreeBN128::getGroupProof (merkleTreeBN128.cpp:210-260) for many indices at once: out[q] = row idx_q (w Goldilocks words)
// followed, per level, by the 16 nodes of the group idx_q falls in (4 words each); one block per query
__global__ __launch_bounds__(64) void bn128_proofs_kernel(u64* __restrict__ out, const u64* __restrict__ src, const u64* __restrict__ nodes, const u64 w, const u64 h,
                                                         const u64* __restrict__ idxs, const u32 levels)
{
    u64 idx = idxs[blockIdx.x];
    u64* o = out + (u64)blockIdx.x * (w + 64ull * levels);
    for (u64 i = threadIdx.x; i < w; i += blockDim.x) o[i] = src[idx * w + i];
    u64 offset = 0, n = h;
    for (u32 l = 0; l < levels; l++) {
        const u64 si = idx & ~15ull;
        o[w + 64ull * l + threadIdx.x] = nodes[(offset + si) * 4 + threadIdx.x];     // 16 nodes x 4 words = 64 lanes
        const u64 next = (n - 1) / 16 + 1;
        offset += next * 16; idx >>= 4; n = next;
    }
}

// ONE permutation spread over the lanes of one wave (lane i owns state element i): the transcript asks for its permutations one at
// a time (TranscriptBN128::_updateState), and a lane walking a whole t = 17 permutation alone needs ~6 ms (5 457 dependent Montgomery
// products).  Here the critical path is 8 x (3 + t) + nP x ~5 + t products: ~0.6 ms.  Same arithmetic, same result.
__global__ __launch_bounds__(WAVE) void bn128_perm_coop_kernel(Fr* __restrict__ state, const u32 t, const Tables K)
{
    __shared__ u32 sh[MAXT * 8];            // state / product vector, element-major
    __shared__ u32 xs[8];                   // broadcast of the S-boxed element 0
    const u32 lane = threadIdx.x;
    const bool on = lane < t;
    const int k = (int)t - 2;
    const Fr* __restrict__ c = K.C[k];
    const Fr* __restrict__ s = K.S[k];
    const u32 nP = K.nP[k];
    auto put = [&](const Fr& e) { if (on) for (int l = 0; l < 8; l++) sh[lane * 8 + l] = e.v[l]; };
    auto get = [&](u32 j) { Fr e; for (int l = 0; l < 8; l++) e.v[l] = sh[j * 8 + l]; return e; };
    auto mixc = [&](const Fr& mine, const Fr* __restrict__ mat) {
        put(mine);
        __syncthreads();
        Fr acc = zero();
        for (u32 j = 0; j < t; j++) acc = add(acc, mul(gld(&mat[j * t + (on ? lane : 0)]), get(j)));
        __syncthreads();
        return acc;
    };
    Fr st = on ? gld(&state[lane]) : zero();
    st = add(st, gld(&c[on ? lane : 0]));
    for (int fr = 0; fr < 8; fr++) {
        if (fr == 4) {
            for (u32 r = 0; r < nP; r++) {
                const u32 base = (2 * t - 1) * r;
                if (lane == 0) {
                    st = add(exp5(st), gld(&c[5 * t + r]));
                    for (int l = 0; l < 8; l++) xs[l] = st.v[l];
                }
                __syncthreads();
                Fr x;
                for (int l = 0; l < 8; l++) x.v[l] = xs[l];
                put(mul(gld(&s[base + (on ? lane : 0)]), st));                       // S[j] * state[j] (lane 0: the S-boxed element)
                if (on && lane > 0) st = add(st, mul(x, gld(&s[base + t + lane - 1])));
                __syncthreads();
                if (lane == 0) {
                    Fr s0 = get(0);
                    for (u32 j = 1; j < t; j++) s0 = add(s0, get(j));
                    st = s0;
                }
                __syncthreads();
            }
        }
        const u32 cidx = fr < 4 ? (u32)((fr + 1) * t) : (u32)(5 * t + nP + (fr - 4) * t);
        st = exp5(st);
        if (fr != 7) st = add(st, gld(&c[cidx + (on ? lane : 0)]));
        st = mixc(st, fr == 3 ? K.P[k] : K.M[k]);
    }
    if (on) gst(&state[lane], st);
}

}  // namespace bn

// ----------------------------------------------------------------- host side
static bn::Tables g_bn_tables;
static void* g_bn_blob = nullptr;
static bool g_bn_ready = false;

static std::string default_constants_path()
{
    Dl_info info;
    if (dladdr((const void*)&default_constants_path, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t k = p.rfind('/');
        return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/chelpers/poseidon_bn128.bin";
    }
    return "chelpers/poseidon_bn128.bin";
}

// the tables live on the device the context was bound to: they go with it (xlp_shutdown), a later xlp_init loads them again
void bn128_shutdown()
{
    if (g_bn_blob) { (void)plan_free(g_bn_blob); g_bn_blob = nullptr; }
    g_bn_tables = bn::Tables{};
    g_bn_ready = false;
}

int bn128_init(Ctx* ctx, const char* path)
{
    (void)ctx;
    if (g_bn_ready && !path) return 0;
    const std::string file = path && *path ? std::string(path) : default_constants_path();
    FILE* f = fopen(file.c_str(), "rb");
    if (!f) return xlp_fail(("bn128: Poseidon-BN128 constants not found: " + file + " (tools/gen_poseidon_bn128.py writes them)").c_str());
    std::vector<unsigned char> raw;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + n);
    fclose(f);
    if (raw.size() < 16 + 32 * 16 || memcmp(raw.data(), "XLPBN128", 8) != 0) return xlp_fail("bn128: bad constants file");
    const u64* cnt = (const u64*)(raw.data() + 16);
    static const u32 nrp[16] = {56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68};   // poseidon_opt.hpp:18
    u64 total = 0;
    for (int k = 0; k < 16; k++) {
        const u64 t = (u64)k + 2;
        if (cnt[4 * k] != 8 * t + nrp[k] || cnt[4 * k + 1] != (2 * t - 1) * nrp[k] || cnt[4 * k + 2] != t * t || cnt[4 * k + 3] != t * t)
            return xlp_fail("bn128: constants file has unexpected table sizes");
        total += cnt[4 * k] + cnt[4 * k + 1] + cnt[4 * k + 2] + cnt[4 * k + 3];
    }
    const size_t body = 16 + 32 * 16;
    if (raw.size() != body + total * 32) return xlp_fail("bn128: constants file truncated");
    if (g_bn_blob) { (void)plan_free(g_bn_blob); g_bn_blob = nullptr; }
    XLP_ALLOC(plan_malloc((void**)&g_bn_blob, total * 32));
    XLP_HIP(hipMemcpy(g_bn_blob, raw.data() + body, total * 32, hipMemcpyHostToDevice));
    const bn::Fr* p = (const bn::Fr*)g_bn_blob;
    for (int k = 0; k < 16; k++) {
        g_bn_tables.C[k] = p; p += cnt[4 * k];
        g_bn_tables.S[k] = p; p += cnt[4 * k + 1];
        g_bn_tables.M[k] = p; p += cnt[4 * k + 2];
        g_bn_tables.P[k] = p; p += cnt[4 * k + 3];
        g_bn_tables.nP[k] = nrp[k];
    }
    g_bn_ready = true;
    return 0;
}

u64 bn128_num_nodes(u64 n)       // MerkleTreeBN128::getNumNodes (merkleTreeBN128.cpp:62-84)
{
    if (n == 0) return 0;
    u64 n_tmp = n;
    u64 next = (n_tmp - 1) / 16 + 1;
    u64 acc = next * 16;
    while (n_tmp > 1) {
        n_tmp = next;
        next = (n_tmp - 1) / 16 + 1;
        acc += n_tmp > 1 ? next * 16 : 1;
    }
    return acc;
}

int dev_merkle_bn128(Ctx* ctx, u64* nodes, const u64* src, u64 w, u64 h)
{
    if (bn128_init(ctx, nullptr)) return -1;
    if (h == 0) return 0;
    if (h & (h - 1)) return xlp_fail("merkle_bn128: height must be a power of two");
    bn::Fr* nd = (bn::Fr*)nodes;
    XLP_HIP(hipMemsetAsync(nd, 0, bn128_num_nodes(h) * sizeof(bn::Fr), ctx->stream));     // calloc'ed in the reference: padding slots stay zero
    const u64 nfr = (w + 2) / 3;          // Fr elements per leaf: a leaf hashes min(nfr, 16) + 1 elements at a time
    if (w > 4 && nfr <= 4) hipLaunchKernelGGL(bn::bn128_leaf_kernel<5>, dim3((unsigned)((h + bn::WAVE - 1) / bn::WAVE)), dim3(bn::WAVE), 0, ctx->stream, nd, src, w, h, g_bn_tables);
    else if (w > 4 && nfr <= 7) hipLaunchKernelGGL(bn::bn128_leaf_kernel<8>, dim3((unsigned)((h + bn::WAVE - 1) / bn::WAVE)), dim3(bn::WAVE), 0, ctx->stream, nd, src, w, h, g_bn_tables);
    else hipLaunchKernelGGL(bn::bn128_leaf_kernel<bn::MAXT>, dim3((unsigned)((h + bn::WAVE - 1) / bn::WAVE)), dim3(bn::WAVE), 0, ctx->stream, nd, src, w, h, g_bn_tables);
    XLP_HIP(hipGetLastError());
    u64 n256 = h, cursor = 0, cursor_next = h;
    u64 next_n = (n256 - 1) / 16 + 1;
    while (n256 > 1) {
        const u64 batches = (n256 + 15) / 16;
        const u32 num = batches == 1 ? (u32)n256 : 16u;
        hipLaunchKernelGGL(bn::bn128_level_kernel, dim3((unsigned)((batches + bn::WAVE - 1) / bn::WAVE)), dim3(bn::WAVE), 0, ctx->stream,
                           nd + cursor_next, nd + cursor, batches, num, g_bn_tables);
        XLP_HIP(hipGetLastError());
        n256 = next_n;
        next_n = (n256 - 1) / 16 + 1;
        cursor = cursor_next;
        cursor_next = cursor + (n256 > 16 ? n256 : 16);
    }
    return 0;
}

int dev_merkle_bn128_proofs(Ctx* ctx, u64* out, const u64* src, const u64* nodes, u64 w, u64 h, const u64* d_idxs, u64 nq)
{
    if (!nq) return 0;
    if (h == 0 || (h & (h - 1))) return xlp_fail("merkle_bn128 proofs: height must be a power of two");
    u32 levels = 0;
    for (u64 c = 1; c < h; c *= 16) levels++;
    hipLaunchKernelGGL(bn::bn128_proofs_kernel, dim3((unsigned)nq), dim3(64), 0, ctx->stream, out, src, nodes, w, h, d_idxs, levels);
    XLP_HIP(hipGetLastError());
    return 0;
}

int dev_poseidon_bn128(Ctx* ctx, u64* states, u32 t, u64 count)
{
    if (bn128_init(ctx, nullptr)) return -1;
    if (t < 2 || t > 17) return xlp_fail("poseidon_bn128: state size must be 2..17");
    if (count == 0) return 0;
    if (count <= 4) {          // the transcript's one-at-a-time calls: one wave per permutation, lanes across the state
        for (u64 q = 0; q < count; q++) hipLaunchKernelGGL(bn::bn128_perm_coop_kernel, dim3(1), dim3(bn::WAVE), 0, ctx->stream, (bn::Fr*)states + q * t, t, g_bn_tables);
        XLP_HIP(hipGetLastError());
        return 0;
    }
    if (t <= 8) hipLaunchKernelGGL(bn::bn128_perm_kernel<8>, dim3((unsigned)((count + bn::WAVE - 1) / bn::WAVE)), dim3(bn::WAVE), 0, ctx->stream, (bn::Fr*)states, t, count, g_bn_tables);
    else hipLaunchKernelGGL(bn::bn128_perm_kernel<bn::MAXT>, dim3((unsigned)((count + bn::WAVE - 1) / bn::WAVE)), dim3(bn::WAVE), 0, ctx->stream, (bn::Fr*)states, t, count, g_bn_tables);
    XLP_HIP(hipGetLastError());
    return 0;
}

// ---- host-side scalar Fr arithmetic for launch constants only (root of unity, 1 / n): 4 x 64-bit Montgomery
namespace hfr {
typedef unsigned __int128 u128;
struct E { u64 v[4]; };
static const u64 Q[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
static const u64 NP = 0xc2e1f593efffffffULL;
static E mont(const E& a, const E& b)
{
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { const u128 p = (u128)a.v[j] * b.v[i] + t[j] + (u64)c; t[j] = (u64)p; c = p >> 64; }
        u128 s = (u128)t[4] + (u64)c; t[4] = (u64)s; t[5] = (u64)(s >> 64);
        const u64 m = t[0] * NP;
        u128 p = (u128)m * Q[0] + t[0]; c = p >> 64;
        for (int j = 1; j < 4; j++) { p = (u128)m * Q[j] + t[j] + (u64)c; t[j - 1] = (u64)p; c = p >> 64; }
        s = (u128)t[4] + (u64)c; t[3] = (u64)s; t[4] = t[5] + (u64)(s >> 64);
    }
    E r = {{t[0], t[1], t[2], t[3]}};
    bool ge = t[4] != 0;
    if (!ge) { ge = true; for (int i = 3; i >= 0; i--) if (r.v[i] != Q[i]) { ge = r.v[i] > Q[i]; break; } }
    if (ge) { u128 bw = 0; for (int i = 0; i < 4; i++) { const u128 d = (u128)r.v[i] - Q[i] - (u64)bw; r.v[i] = (u64)d; bw = (d >> 64) & 1; } }
    return r;
}
static E from_u64(u64 x) { E raw = {{x, 0, 0, 0}}, r2 = {{R2[0], R2[1], R2[2], R2[3]}}; return mont(raw, r2); }
static E pow(E base, const u64 e[4])
{
    E r = from_u64(1);
    for (int i = 255; i >= 0; i--) {
        r = mont(r, r);
        if ((e[i >> 6] >> (i & 63)) & 1) r = mont(r, base);
    }
    return r;
}
}  // namespace hfr

static bn::Fr to_dev(const hfr::E& e)
{
    bn::Fr f;
    for (int i = 0; i < 4; i++) { f.v[2 * i] = (u32)e.v[i]; f.v[2 * i + 1] = (u32)(e.v[i] >> 32); }
    return f;
}

// FFT<RawFr>::fft / ifft (src/ffiasm/fft.c.hpp:181-221) of n = 2^logn Fr elements (Montgomery limbs) in place on the device:
// w_n = nqr^((q - 1) / n) with nqr = 5, the smallest quadratic non-residue (fft.c.hpp:52-57,83-84)
int dev_fft_bn254(Ctx* ctx, u64* data, u64 n, int inverse)
{
    if (n == 0) return 0;
    if (n & (n - 1)) return xlp_fail("fft_bn254: size must be a power of two");
    u32 logn = 0;
    while ((1ull << logn) < n) logn++;
    if (logn > 28) return xlp_fail("fft_bn254: domain too big for the curve (2-adicity 28)");
    bn::Fr* a = (bn::Fr*)data;
    if (logn == 0) return 0;
    // The table of twiddles w^j (j < n / 2) lives in a scratch slot of its own and is kept from call to call: a table built for a larger domain
    // serves a smaller one (root(s, j) = w_L^(j << (L - s))).  (Until round 4 every call rebuilt it: 0.9 ms of a 5.7 ms transform at 2^24.)
    u32 tl = ctx->fft_tw_log >= logn ? ctx->fft_tw_log : logn;
    u64* tslot = nullptr;
    if (ctx_scratch(ctx, ((1ull << (tl - 1)) + 32) * sizeof(bn::Fr), 8, &tslot)) return -1;
    bn::Fr* tw = (bn::Fr*)tslot;
    if (ctx->fft_tw_log < logn || ctx->fft_tw_ptr != (const void*)tslot) {
        tl = logn;
        const u64 half_t = 1ull << (tl - 1);
        u64 e[4];                                   // (q - 1) >> tl
        {
            u64 qm1[4] = {hfr::Q[0] - 1, hfr::Q[1], hfr::Q[2], hfr::Q[3]};
            for (int i = 0; i < 4; i++) e[i] = (qm1[i] >> tl) | (i < 3 && tl ? qm1[i + 1] << (64 - tl) : 0);
        }
        hfr::E w = hfr::pow(hfr::from_u64(5), e);
        std::vector<bn::Fr> wpow(tl);
        for (u32 k = 0; k < tl; k++) { wpow[k] = to_dev(w); w = hfr::mont(w, w); }
        bn::Fr* d_wpow = tw + half_t;
        XLP_HIP(hipMemcpyAsync(d_wpow, wpow.data(), tl * sizeof(bn::Fr), hipMemcpyHostToDevice, ctx->stream));
        XLP_HIP(hipStreamSynchronize(ctx->stream));       // wpow is a host temporary
        hipLaunchKernelGGL(bn::fr_twiddle_kernel, dim3((unsigned)((half_t + 255) / 256)), dim3(256), 0, ctx->stream, tw, half_t, d_wpow, to_dev(hfr::from_u64(1)));
        XLP_HIP(hipGetLastError());
        ctx->fft_tw_log = tl;
        ctx->fft_tw_ptr = (const void*)tslot;
    }
    const u64 half = n >> 1;
    hipLaunchKernelGGL(bn::fr_bitrev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, logn);
    const u32 nst = logn < (u32)bn::FFT_LB ? logn : (u32)bn::FFT_LB;
    hipLaunchKernelGGL(bn::fr_fft_low_kernel, dim3((unsigned)(n >> nst)), dim3(1 << (bn::FFT_LB - 1)), 0, ctx->stream, a, tw, tl, nst);
    // the stages above the first LDS pass, up to 8 at a time (fr_fft_mid_kernel); s0 = stages done so far
    for (u32 s0 = nst; s0 < logn;) {
        const u32 m = logn - s0 < 8 ? logn - s0 : 8;                       // stages of this pass
        u32 lt = (u32)bn::FFT_MB - m;                                       // log2 T: the tile is 2^FFT_MB elements ...
        if (lt > s0) lt = s0;                                               // ... when that many consecutive lo exist
        hipLaunchKernelGGL(bn::fr_fft_mid_kernel, dim3((unsigned)(n >> (m + lt))), dim3(256), 0, ctx->stream, a, tw, s0, m, lt, tl);
        s0 += m;
    }
    if (inverse) {
        // 1 / n = 2^-logn: (q + 1) / 2 raised to logn
        hfr::E inv2 = hfr::from_u64(2);
        u64 qm2[4] = {hfr::Q[0] - 2, hfr::Q[1], hfr::Q[2], hfr::Q[3]};
        inv2 = hfr::pow(inv2, qm2);
        hfr::E ninv = hfr::from_u64(1);
        for (u32 k = 0; k < logn; k++) ninv = hfr::mont(ninv, inv2);
        hipLaunchKernelGGL(bn::fr_ifft_finish_kernel, dim3((unsigned)((half + 1 + 255) / 256)), dim3(256), 0, ctx->stream, a, logn, to_dev(ninv));
    }
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
