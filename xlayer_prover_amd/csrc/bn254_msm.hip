// G1 multi-scalar multiplication over BN254 (alt_bn128) for gfx950 -- the commitment primitive of the fflonk prover
// (SURVEY.md 8 f4): r = sum_i scalars[i] * bases[i].
//
// Replaces  ParallelMultiexp<Curve>::multiexp   src/ffiasm/multiexp.c.hpp:118-164  (as called through
//           Curve::multiMulByScalar, src/ffiasm/curve.c.hpp, from src/rapidsnark/fflonk_prover.c.hpp:1700,1712)
// Semantics kept: bases are affine points (x, y) in Fq Montgomery form, (0, 0) = the point at infinity and is skipped
// (multiexp.c.hpp:41); scalars are raw little-endian integers of 32 bytes (NOT reduced, NOT Montgomery); the result is the
// group element, returned in affine form (the reference returns a projective representative whose coordinates depend on the
// thread count; callers normalise it).
//
// Design (not the reference's per-thread bucket arrays).  Bucket method with c-bit windows, all windows at once:
//   1. msm_keys      (window w, point i) -> key = w * 2^c + digit, value = i             one pass over the scalars
//   2. radix sort of the (key, value) pairs (rocPRIM through hipCUB), bucket start offsets by neighbour comparison
//   3. msm_buckets   one lane per non-empty-able bucket (w, digit != 0): Jacobian accumulator += its points (mixed addition)
//   4. msm_groups    per window, 2^c / 64 lanes: running-sum trick on 64 consecutive buckets, then the group's offset * total;
//      msm_window_sums: one workgroup per window folds the groups (LDS tree)
//   5. on the host: Horner over the windows (c doublings each), Jacobian -> affine (hostfq::finish; the call synchronises the stream)
// Fq arithmetic: 8 x 32-bit limbs, Montgomery (R = 2^256), CIOS on v_mad_u64_u32 -- integer-VALU bound like everything else here.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include "xlp_internal.hpp"
#include "mont256.hpp"

namespace xlp {
namespace g1 {

struct Fq { u32 v[8]; };
// q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
__device__ __constant__ u32 Q_MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
__device__ __constant__ u32 Q_ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};   // R mod q
constexpr u32 Q_N0INV = 0xe4866389u;

__device__ __forceinline__ Fq fzero() { Fq z; for (int i = 0; i < 8; i++) z.v[i] = 0; return z; }
__device__ __forceinline__ Fq fone() { Fq z; for (int i = 0; i < 8; i++) z.v[i] = Q_ONE[i]; return z; }
__device__ __forceinline__ bool is_zero(const Fq& a) { u32 o = 0; for (int i = 0; i < 8; i++) o |= a.v[i]; return o == 0; }
__device__ __forceinline__ bool feq(const Fq& a, const Fq& b) { u32 o = 0; for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i]; return o == 0; }

__device__ __forceinline__ Fq cond_sub(const u32 (&a)[8], u32 hi)
{
    u32 d[8];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 s = (u64)a[i] - Q_MOD[i] - borrow;
        d[i] = (u32)s;
        borrow = (u32)(s >> 63);
    }
    const bool ge = hi != 0 || borrow == 0;
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = ge ? d[i] : a[i];
    return r;
}
__device__ __forceinline__ Fq fadd(const Fq& a, const Fq& b)
{
    u32 s[8];
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 t = (u64)a.v[i] + b.v[i] + c;
        s[i] = (u32)t;
        c = (u32)(t >> 32);
    }
    return cond_sub(s, c);
}
__device__ __forceinline__ Fq fsub(const Fq& a, const Fq& b)
{
    u32 d[8];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 s = (u64)a.v[i] - b.v[i] - borrow;
        d[i] = (u32)s;
        borrow = (u32)(s >> 63);
    }
    u32 c = 0;
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 t = (u64)d[i] + (borrow ? Q_MOD[i] : 0u) + c;
        r.v[i] = (u32)t;
        c = (u32)(t >> 32);
    }
    return r;
}
__device__ __forceinline__ Fq fdbl(const Fq& a) { return fadd(a, a); }
struct FqParams {
    static constexpr u32 N[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr u32 N0INV = 0xe4866389u;
};
// (operands BY VALUE: 2 x 8 registers in, 8 out.  By reference the caller parks both operands in scratch memory and the function starts with
// four loads and a wait -- a memory round trip in front of each of the 11 products of a mixed addition)
__device__ __noinline__ Fq fmul(const Fq a, const Fq b)        // mont256.hpp: product scanning, 2 VALU per multiply-accumulate
{
    Fq r;
    mont::mul<FqParams>(r.v, a.v, b.v);
    return r;
}
__device__ __forceinline__ Fq fsqr(const Fq& a) { return fmul(a, a); }
// the product inlined: for the one addition formula that runs per (point, window) pair (jmadd_inl; 38 KB of kernel, fflonk 2^22 -1.7 % against calls)
__device__ __forceinline__ Fq fmul_i(const Fq& a, const Fq& b) { Fq r; mont::mul<FqParams>(r.v, a.v, b.v); return r; }
__device__ __forceinline__ Fq fsqr_i(const Fq& a) { return fmul_i(a, a); }

struct Aff { Fq x, y; };            // (0, 0) = infinity
struct Jac { Fq x, y, z; };         // z = 0 = infinity

__device__ __forceinline__ Jac jzero() { Jac p; p.x = fone(); p.y = fone(); p.z = fzero(); return p; }
__device__ __forceinline__ bool jinf(const Jac& p) { return is_zero(p.z); }
__device__ __forceinline__ bool ainf(const Aff& p) { return is_zero(p.x) && is_zero(p.y); }

// dbl-2009-l (a = 0).  jdbl_inl: for the rare equal-points branch of jmadd_inl -- as a CALL there, the callee's register budget became the bucket
// kernels' (196 registers, 2 wavefronts per SIMD; inlined: 128, 4 wavefronts)
__device__ __forceinline__ Jac jdbl_inl(const Jac& p)
{
    if (jinf(p)) return p;
    const Fq A = fsqr(p.x), B = fsqr(p.y), C = fsqr(B);
    Fq D = fsub(fsub(fsqr(fadd(p.x, B)), A), C);
    D = fdbl(D);
    const Fq E = fadd(fdbl(A), A), F = fsqr(E);
    Jac r;
    r.x = fsub(F, fdbl(D));
    Fq C8 = fdbl(fdbl(fdbl(C)));
    r.y = fsub(fmul(E, fsub(D, r.x)), C8);
    r.z = fdbl(fmul(p.y, p.z));
    return r;
}
__device__ __noinline__ Jac jdbl(const Jac& p) { return jdbl_inl(p); }
// madd-2007-bl with the exceptional cases.  jmadd_inl: inlined into the loops of the bucket kernels -- the accumulator and the temporaries
// then live in the KERNEL's registers; as a function of its own it saves and restores ~100 callee-saved registers
// through scratch memory on every call (464 bytes of stack per lane) and hands its result back through memory.
__device__ __forceinline__ Jac jmadd_inl(const Jac& p, const Aff& q)
{
    if (ainf(q)) return p;
    if (jinf(p)) { Jac r; r.x = q.x; r.y = q.y; r.z = fone(); return r; }
    const Fq Z1Z1 = fsqr_i(p.z), U2 = fmul_i(q.x, Z1Z1), S2 = fmul_i(fmul_i(q.y, p.z), Z1Z1);
    const Fq H = fsub(U2, p.x);
    Fq rr = fsub(S2, p.y);
    if (is_zero(H)) {
        if (is_zero(rr)) { Jac t; t.x = q.x; t.y = q.y; t.z = fone(); return jdbl_inl(t); }
        return jzero();
    }
    rr = fdbl(rr);
    const Fq HH = fsqr_i(H), I = fdbl(fdbl(HH)), J = fmul_i(H, I), V = fmul_i(p.x, I);
    Jac r;
    r.x = fsub(fsub(fsqr_i(rr), J), fdbl(V));
    r.y = fsub(fmul_i(rr, fsub(V, r.x)), fdbl(fmul_i(p.y, J)));
    r.z = fsub(fsub(fsqr_i(fadd(p.z, H)), Z1Z1), HH);
    return r;
}
__device__ __noinline__ Jac jmadd(const Jac& p, const Aff& q) { return jmadd_inl(p, q); }
// add-2007-bl with the exceptional cases (jadd_inl: for the running sums of msm_groups_kernel, as jmadd_inl above)
__device__ __forceinline__ Jac jadd_inl(const Jac& p, const Jac& q)
{
    if (jinf(q)) return p;
    if (jinf(p)) return q;
    const Fq Z1Z1 = fsqr(p.z), Z2Z2 = fsqr(q.z);
    const Fq U1 = fmul(p.x, Z2Z2), U2 = fmul(q.x, Z1Z1);
    const Fq S1 = fmul(fmul(p.y, q.z), Z2Z2), S2 = fmul(fmul(q.y, p.z), Z1Z1);
    const Fq H = fsub(U2, U1);
    Fq rr = fsub(S2, S1);
    if (is_zero(H)) {
        if (is_zero(rr)) return jdbl_inl(p);
        return jzero();
    }
    rr = fdbl(rr);
    const Fq I = fsqr(fdbl(H)), J = fmul(H, I), V = fmul(U1, I);
    Jac r;
    r.x = fsub(fsub(fsqr(rr), J), fdbl(V));
    r.y = fsub(fmul(rr, fsub(V, r.x)), fdbl(fmul(S1, J)));
    r.z = fmul(fsub(fsub(fsqr(fadd(p.z, q.z)), Z1Z1), Z2Z2), H);
    return r;
}
__device__ __noinline__ Jac jadd(const Jac& p, const Jac& q) { return jadd_inl(p, q); }
// k * p for a small k (<= 16 bits), double-and-add from the top
__device__ __forceinline__ Jac jmul_small(const Jac& p, u32 k)
{
    Jac r = jzero();
    for (int b = 31 - __clz(k | 1); b >= 0; b--) {
        r = jdbl_inl(r);
        if ((k >> b) & 1) r = jadd_inl(r, p);
    }
    return k ? r : jzero();
}

// ----------------------------------------------------------------- kernels
// digit of scalar i in window w: bits [w * c, w * c + c) of the 256-bit little-endian integer (getChunk, multiexp.c.hpp:23-35)
__device__ __forceinline__ u32 window_digit(const u32* __restrict__ sc, u32 w, u32 c)
{
    const u32 bit = w * c, word = bit >> 5, sh = bit & 31;
    u64 v = sc[word];
    if (word + 1 < 8) v |= (u64)sc[word + 1] << 32;
    return (u32)((v >> sh) & ((1u << c) - 1));
}

// one lane per point: its scalar and its base are read once, the nwin (key, value) pairs of its windows written window by window
// (until round 4 one lane per (window, point): 13 reads of the 64-byte base just to test it for infinity, 9 ms of a 135 ms sum)
__global__ void msm_keys_kernel(u32* __restrict__ keys, u32* __restrict__ vals, const u32* __restrict__ scalars, const Aff* __restrict__ bases,
                                const u64 n, const u32 nwin, const u32 c)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool inf = ainf(bases[i]);                 // points at infinity are skipped: digit 0 is the "no contribution" bucket
    const u32* __restrict__ sc = scalars + i * 8;     // (32 bytes: one cache line serves all windows)
    for (u32 w = 0; w < nwin; w++) {
        const u32 d = inf ? 0u : window_digit(sc, w, c);
        keys[(u64)w * n + i] = (w << c) | d;
        vals[(u64)w * n + i] = (u32)i;
    }
}

// start[k] / stop[k] = first / one past the last sorted position with key k; a key absent from the sorted array keeps start = stop = 0
// (both arrays are zeroed first).  (Until round 4 absent keys inherited the next key's start: the lane at the END of the array then filled
// the whole unused top of the top window -- up to 2^c words, one after the other: 30 ms of a 195 ms sum at 9 x 2^22 points.)
__global__ void msm_starts_kernel(u32* __restrict__ start, u32* __restrict__ stop, const u32* __restrict__ sorted_keys, const u64 total)
{
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const u32 cur = sorted_keys[gid];
    if (gid == 0 || sorted_keys[gid - 1] != cur) start[cur] = (u32)gid;
    if (gid + 1 == total || sorted_keys[gid + 1] != cur) stop[cur] = (u32)(gid + 1);
}

// A bucket with more than HEAVY points is not walked by one lane: skewed scalars (many equal digits) and the top window (254 mod c
// significant bits: few buckets, n / 2^bits points each) would serialise thousands of additions on one lane.  Such buckets are listed and
// summed in SEGMENTS of SEG points, one workgroup per segment (msm_heavy_kernel), the partial sums of a bucket folded by one workgroup
// (msm_heavy_fold_kernel): a bucket of millions of points (round 4: a 21-bit window leaves 2 bits to the top window, three buckets of
// 9 x 2^22 / 4 points each -- 1 s of three lonely workgroups) spreads over the chip.
// heavy[0] = segments so far, heavy[1] = heavy buckets so far; hb[3 i ..] = (key, first segment, segments); hs[2 e ..] = (key, segment)
constexpr u32 HEAVY = 256;
constexpr u32 SEG = 16384;     // (64 additions per lane before the 8-level tree of full additions; 4 096: the same wall time at 9 x 2^22 points)
// cnt[key] = points of the bucket, capped (the sort key that puts buckets of like size into the same wavefront)
__global__ void msm_counts_kernel(u32* __restrict__ cnt, u32* __restrict__ ids, const u32* __restrict__ start, const u32* __restrict__ stop, const u32 nkeys, const u32 c)
{
    const u32 key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= nkeys) return;
    const u32 n = (key & ((1u << c) - 1)) != 0 ? stop[key] - start[key] : 0;
    cnt[key] = n < 511 ? n : 511;
    ids[key] = key;
}
// One lane per bucket, the buckets taken in order of their size (`order`): the 64 lanes of a wavefront walk runs of (nearly) the same length.
// In bucket order a wavefront waits for its longest run -- Poisson(16) sizes: 16 of 27 lane-steps useful (round 4: 109 -> 70 ms at 9 x 2^22 points).
__global__ __launch_bounds__(64) void msm_buckets_kernel(Jac* __restrict__ buckets, const u32* __restrict__ start, const u32* __restrict__ stop, const u32* __restrict__ sorted_vals,
                                                          const Aff* __restrict__ bases, const u32 nkeys, const u32 c, u32* __restrict__ heavy, u32* __restrict__ hb, u32* __restrict__ hs,
                                                          const u32* __restrict__ order)
{
    const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nkeys) return;
    const u32 key = order[gid];
    Jac acc = jzero();
    if ((key & ((1u << c) - 1)) != 0) {
        const u32 b = start[key], e = stop[key];
        if (e - b > HEAVY) {
            const u32 nseg = (e - b + SEG - 1) / SEG;
            const u32 first = atomicAdd(&heavy[0], nseg), i = atomicAdd(&heavy[1], 1u);
            hb[3 * i] = key; hb[3 * i + 1] = first; hb[3 * i + 2] = nseg;
            for (u32 s_ = 0; s_ < nseg; s_++) { hs[2 * (first + s_)] = key; hs[2 * (first + s_) + 1] = s_; }
            return;
        }
        for (u32 k = b; k < e; k++) acc = jmadd_inl(acc, bases[sorted_vals[k]]);
    }
    buckets[key] = acc;
}
__device__ __forceinline__ Jac block_sum_256(Jac acc, Jac* sh, const u32 t)
{
    sh[t] = acc;
    __syncthreads();
    for (u32 step = 128; step >= 1; step >>= 1) {
        if (t < step) sh[t] = jadd_inl(sh[t], sh[t + step]);
        __syncthreads();
    }
    const Jac r = sh[0];
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(256) void msm_heavy_kernel(Jac* __restrict__ part, const u32* __restrict__ start, const u32* __restrict__ stop, const u32* __restrict__ sorted_vals,
                                                         const Aff* __restrict__ bases, const u32* __restrict__ heavy, const u32* __restrict__ hs)
{
    __shared__ Jac sh[256];
    const u32 t = threadIdx.x;
    for (u32 e_ = blockIdx.x; e_ < heavy[0]; e_ += gridDim.x) {
        const u32 key = hs[2 * e_], seg = hs[2 * e_ + 1];
        const u32 b = start[key] + seg * SEG, lim = stop[key], e = b + SEG < lim ? b + SEG : lim;
        Jac acc = jzero();
        for (u32 k = b + t; k < e; k += 256) acc = jmadd_inl(acc, bases[sorted_vals[k]]);
        const Jac r = block_sum_256(acc, sh, t);
        if (t == 0) part[e_] = r;
    }
}
__global__ __launch_bounds__(256) void msm_heavy_fold_kernel(Jac* __restrict__ buckets, const Jac* __restrict__ part, const u32* __restrict__ heavy, const u32* __restrict__ hb)
{
    __shared__ Jac sh[256];
    const u32 t = threadIdx.x;
    for (u32 i = blockIdx.x; i < heavy[1]; i += gridDim.x) {
        const u32 key = hb[3 * i], first = hb[3 * i + 1], nseg = hb[3 * i + 2];
        Jac acc = jzero();
        for (u32 s_ = t; s_ < nseg; s_ += 256) acc = jadd(acc, part[first + s_]);
        const Jac r = block_sum_256(acc, sh, t);
        if (t == 0) buckets[key] = r;
    }
}

// group g of window w covers digits [g * GS, (g + 1) * GS): out = sum_d d * B_d over the group
constexpr u32 GS = 64;       // buckets per running-sum lane: 2^c / 64 lanes per window (a 256-bucket lane was 13 ms of serial additions)
__global__ __launch_bounds__(64) void msm_groups_kernel(Jac* __restrict__ groups, const Jac* __restrict__ buckets, const u32 nwin, const u32 c)
{
    const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nb = 1u << c, gs = nb < GS ? nb : GS, gpw = nb / gs;
    if (gid >= nwin * gpw) return;
    const u32 w = gid / gpw, g = gid % gpw;
    const Jac* __restrict__ B = buckets + ((u64)w << c) + (u64)g * gs;
    Jac run = jzero(), sum = jzero();
    for (int d = (int)gs - 1; d >= 0; d--) {          // sum = sum_d (d + 1) * B[d] ... shifted below
        run = jadd_inl(run, B[d]);
        sum = jadd_inl(sum, run);
    }
    // sum = sum_d (d + 1) B[d]; wanted sum_d (g * gs + d) B[d] = sum + (g * gs - 1) * run  (run = group total); g * gs = 0: sum - run
    if (g == 0) {
        Jac neg = run; neg.y = fsub(fzero(), neg.y);
        groups[gid] = jadd_inl(sum, neg);
    } else {
        groups[gid] = jadd_inl(sum, jmul_small(run, g * gs - 1));
    }
}

// per-window sum of the groups: one 256-lane workgroup per window, every lane folds its share, then an LDS tree
__global__ __launch_bounds__(256) void msm_window_sums_kernel(Jac* __restrict__ wsum, const Jac* __restrict__ groups, const u32 nwin, const u32 gpw)
{
    __shared__ Jac sh[256];
    const u32 w = blockIdx.x, t = threadIdx.x;
    Jac s = jzero();
    for (u32 g = t; g < gpw; g += 256) s = jadd(s, groups[w * gpw + g]);
    sh[t] = s;
    __syncthreads();
    for (u32 step = 128; step >= 1; step >>= 1) {
        if (t < step) sh[t] = jadd(sh[t], sh[t + step]);
        __syncthreads();
    }
    if (t == 0) wsum[w] = sh[0];
}

// MEASUREMENT AID: the bucket pass's arithmetic without its gathers -- every lane of a 64-lane workgroup (msm_buckets_kernel's launch
// bounds) adds ONE affine point to its register-resident accumulator `iters` times, the additions chained as in a bucket's walk; the point is
// read where it lies, as in the pass (jmadd takes it by reference: a copy in registers would be spilled to scratch for the call), but it is
// the same 64 bytes for every lane and step -- a cache hit, no HBM traffic.  What this reaches in mixed additions per second is the ceiling
// of THIS formulation (Jacobian + affine, mont256 products) on this chip; the multi-exponentiation's distance from it is what its gathers,
// its sort and its reductions cost.
__global__ __launch_bounds__(64) void msm_resident_kernel(Jac* __restrict__ out, const Aff* __restrict__ point, const u32 lanes, const u32 iters)
{
    const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= lanes) return;
    Jac acc; acc.x = point[0].x; acc.y = point[0].y; acc.z = fone();      // q; the first addition below doubles, none after it meets q again
#pragma unroll 1
    for (u32 k = 0; k < iters; k++) acc = jmadd_inl(acc, point[0]);       // (iters + 1) q
    out[gid] = acc;
}

}  // namespace g1


// The last step on the HOST: Horner over the <= 128 window sums (c doublings each, multiexp.c.hpp:155-159) and the conversion to affine are ~2 300
// field products in ONE dependent chain -- 3.3 ms on one GPU lane (1.5 us per product), 0.15 ms on a host core.
namespace hostfq {
typedef unsigned __int128 u128;
struct F { u64 v[4]; };
struct J { F x, y, z; };
static const u64 Q[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 NP = 0x87d20782e4866389ULL;                       // -q^-1 mod 2^64
static inline bool geq(const u64* t) { for (int i = 3; i >= 0; i--) if (t[i] != Q[i]) return t[i] > Q[i]; return true; }
static inline void subq(u64* t) { u128 bw = 0; for (int i = 0; i < 4; i++) { const u128 d = (u128)t[i] - Q[i] - (u64)bw; t[i] = (u64)d; bw = (d >> 64) & 1; } }
static F mul(const F& a, const F& b)                                // Montgomery product, R = 2^256 (the device's representation)
{
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { const u128 p = (u128)a.v[j] * b.v[i] + t[j] + (u64)c; t[j] = (u64)p; c = p >> 64; }
        u128 s_ = (u128)t[4] + (u64)c; t[4] = (u64)s_; t[5] = (u64)(s_ >> 64);
        const u64 m = t[0] * NP;
        u128 p = (u128)m * Q[0] + t[0]; c = p >> 64;
        for (int j = 1; j < 4; j++) { p = (u128)m * Q[j] + t[j] + (u64)c; t[j - 1] = (u64)p; c = p >> 64; }
        s_ = (u128)t[4] + (u64)c; t[3] = (u64)s_; t[4] = t[5] + (u64)(s_ >> 64);
    }
    if (t[4] != 0 || geq(t)) subq(t);
    return F{{t[0], t[1], t[2], t[3]}};
}
static F add(const F& a, const F& b)
{
    u64 t[4]; u128 c = 0;
    for (int i = 0; i < 4; i++) { const u128 s_ = (u128)a.v[i] + b.v[i] + (u64)c; t[i] = (u64)s_; c = s_ >> 64; }
    if (c || geq(t)) subq(t);
    return F{{t[0], t[1], t[2], t[3]}};
}
static F sub(const F& a, const F& b)
{
    u64 t[4]; u128 bw = 0;
    for (int i = 0; i < 4; i++) { const u128 d = (u128)a.v[i] - b.v[i] - (u64)bw; t[i] = (u64)d; bw = (d >> 64) & 1; }
    if (bw) { u128 c = 0; for (int i = 0; i < 4; i++) { const u128 s_ = (u128)t[i] + Q[i] + (u64)c; t[i] = (u64)s_; c = s_ >> 64; } }
    return F{{t[0], t[1], t[2], t[3]}};
}
static inline F dbl(const F& a) { return add(a, a); }
static inline F sqr(const F& a) { return mul(a, a); }
static inline bool zero(const F& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static F one() { F x = {{1, 0, 0, 0}}; for (int i = 0; i < 256; i++) x = dbl(x); return x; }      // 2^256 mod q
static J jdbl(const J& p)                                           // dbl-2009-l, as on the device
{
    if (zero(p.z)) return p;
    const F A = sqr(p.x), B = sqr(p.y), C = sqr(B);
    const F D = dbl(sub(sub(sqr(add(p.x, B)), A), C));
    const F E = add(dbl(A), A), Fv = sqr(E);
    J r;
    r.x = sub(Fv, dbl(D));
    r.y = sub(mul(E, sub(D, r.x)), dbl(dbl(dbl(C))));
    r.z = dbl(mul(p.y, p.z));
    return r;
}
static J jadd(const J& p, const J& q)                               // add-2007-bl with the exceptional cases, as on the device
{
    if (zero(q.z)) return p;
    if (zero(p.z)) return q;
    const F Z1Z1 = sqr(p.z), Z2Z2 = sqr(q.z);
    const F U1 = mul(p.x, Z2Z2), U2 = mul(q.x, Z1Z1);
    const F S1 = mul(mul(p.y, q.z), Z2Z2), S2 = mul(mul(q.y, p.z), Z1Z1);
    const F H = sub(U2, U1);
    F rr = sub(S2, S1);
    if (zero(H)) {
        if (zero(rr)) return jdbl(p);
        J inf; inf.x = one(); inf.y = inf.x; inf.z = F{{0, 0, 0, 0}};
        return inf;
    }
    rr = dbl(rr);
    const F I = sqr(dbl(H)), Jv = mul(H, I), V = mul(U1, I);
    J r;
    r.x = sub(sub(sqr(rr), Jv), dbl(V));
    r.y = sub(mul(rr, sub(V, r.x)), dbl(mul(S1, Jv)));
    r.z = mul(sub(sub(sqr(add(p.z, q.z)), Z1Z1), Z2Z2), H);
    return r;
}
static F inv(const F& a)                                            // a^(q - 2)
{
    const u64 e[4] = {Q[0] - 2, Q[1], Q[2], Q[3]};
    F r = one();
    for (int i = 255; i >= 0; i--) { r = sqr(r); if ((e[i >> 6] >> (i & 63)) & 1) r = mul(r, a); }
    return r;
}
// out[8] (affine, Montgomery; (0, 0) = infinity) = sum_w 2^(c w) wsum[w]
static void finish(u64* out, const J* wsum, u32 nwin, u32 c)
{
    J r; r.x = one(); r.y = r.x; r.z = F{{0, 0, 0, 0}};
    for (int w = (int)nwin - 1; w >= 0; w--) {
        for (u32 k = 0; k < c; k++) r = jdbl(r);
        r = jadd(r, wsum[w]);
    }
    if (zero(r.z)) { for (int i = 0; i < 8; i++) out[i] = 0; return; }
    const F zi = inv(r.z), zi2 = sqr(zi);
    const F x = mul(r.x, zi2), y = mul(r.y, mul(zi2, zi));
    for (int i = 0; i < 4; i++) { out[i] = x.v[i]; out[4 + i] = y.v[i]; }
}
}  // namespace hostfq

// window width for n points (see the comment inside): shared by dev_msm_bn254 and the measurement aid below
static u32 msm_window_bits(u64 n)
{
    u32 lg = 0;
    while ((2ull << lg) <= n) lg++;
    // window width: the bucket pass costs one mixed addition per (point, non-zero digit), the reduction of a window two full additions per
    // bucket (~5 mixed additions' worth of instructions: 14 000 against 2 880, `--pmc SQ_INSTS_VALU`), the sort ~0.2 per pair; scalars have 254
    // significant bits, so the top window often has few buckets.  Pick the width with the least modelled work (round 4: lg - 4, "16 points per
    // bucket", before: at 9 x 2^22 points the model prefers 20 bits to 21)
    u32 c = 2;
    {
        double best = 0;
        const u32 lo = lg > 9 ? lg - 8 : 2, hi = lg < 22 ? (lg > 2 ? lg : 2) : 22;
        for (u32 cc = lo; cc <= hi; cc++) {
            const u32 nw = (256 + cc - 1) / cc;
            if ((double)n * nw >= 2147483648.0) continue;
            double cost = 0;
            for (u32 w = 0; w < nw; w++) {
                const int bits = 254 - (int)(w * cc) >= (int)cc ? (int)cc : 254 - (int)(w * cc);
                if (bits <= 0) { cost += 0.2 * (double)n; continue; }
                const double nb_ = (double)(1ull << bits);
                cost += (double)n * (1.0 - 1.0 / nb_) * 1.0 + 0.2 * (double)n + 4.9 * nb_;
            }
            if (best == 0 || cost < best) { best = cost; c = cc; }
        }
    }
    return c;
}

// out (device, 8 words: x, y affine Montgomery; (0, 0) = infinity) = sum scalars[i] * bases[i]
int dev_msm_bn254(Ctx* ctx, u64* out, const u64* bases, const u64* scalars, u64 n)
{
    using namespace g1;
    if (n == 0) { XLP_HIP(hipMemsetAsync(out, 0, sizeof(Aff), ctx->stream)); return 0; }
    if (n >= (1ull << 28)) return xlp_fail("msm_bn254: more than 2^28 points");
    u32 c = msm_window_bits(n);
    if (const char* e = getenv("XLP_MSM_WINDOW_BITS")) { const int v = atoi(e); if (v >= 2 && v <= 22) c = (u32)v; }     // (experiments)
    const u32 nwin = (256 + c - 1) / c;
    const u64 total = n * nwin;
    if (total >= (1ull << 31)) return xlp_fail("msm_bn254: n * windows exceeds 2^31 pairs");
    const u32 nkeys = nwin << c;
    const u32 nb = 1u << c, gs = nb < GS ? nb : GS, gpw = nb / gs;
    // scratch: keys/vals in+out (4 x total u32), starts (nkeys + 1), buckets (nkeys Jac), groups, window sums, sort temp
    size_t temp_bytes = 0;
    u32* nul = nullptr;
    XLP_HIPV(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, nul, nul, nul, nul, (int)total, 0, 32, ctx->stream));
    const size_t nheavy_max = total / HEAVY + 2;                    // heavy buckets at most
    const size_t nseg_max = total / SEG + nheavy_max + 2;            // their segments at most
    const size_t words32 = 4 * total + 2 * ((size_t)nkeys + 2) + 2 + 3 * nheavy_max + 2 * nseg_max + 4 * (size_t)nkeys;
    const size_t jac_count = (size_t)nkeys + (size_t)nwin * gpw + nwin + nseg_max;
    const size_t bytes = ((words32 * 4 + 255) / 256) * 256 + jac_count * sizeof(Jac) + temp_bytes + 256;
    u64* scratch = nullptr;
    if (ctx_scratch(ctx, bytes, 7, &scratch)) return -1;
    u32* keys = (u32*)scratch;
    u32* vals = keys + total;
    u32* keys2 = vals + total;
    u32* vals2 = keys2 + total;
    u32* start = vals2 + total;
    u32* stop = start + (nkeys + 2);
    u32* heavy = stop + (nkeys + 2);
    u32* hb = heavy + 2;
    u32* hs = hb + 3 * nheavy_max;
    u32* ord = hs + 2 * nseg_max;                                     // bucket sizes / ids, in and out of their sort
    Jac* buckets = (Jac*)((char*)scratch + ((words32 * 4 + 255) / 256) * 256);
    Jac* groups = buckets + nkeys;
    Jac* wsum = groups + (size_t)nwin * gpw;
    Jac* part = wsum + nwin;
    void* temp = (void*)(part + nseg_max);
    const unsigned T = 256;
    hipLaunchKernelGGL(msm_keys_kernel, dim3((unsigned)((n + T - 1) / T)), dim3(T), 0, ctx->stream, keys, vals, (const u32*)scalars, (const Aff*)bases, n, nwin, c);
    u32 key_bits = 0;
    while ((1ull << key_bits) < nkeys) key_bits++;
    XLP_HIP(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys2, vals, vals2, (int)total, 0, (int)key_bits, ctx->stream));
    XLP_HIP(hipMemsetAsync(start, 0, 2 * ((size_t)nkeys + 2) * sizeof(u32), ctx->stream));
    hipLaunchKernelGGL(msm_starts_kernel, dim3((unsigned)((total + T - 1) / T)), dim3(T), 0, ctx->stream, start, stop, keys2, total);
    XLP_HIP(hipMemsetAsync(heavy, 0, 2 * sizeof(u32), ctx->stream));
    // buckets by size
    u32 *cnt = ord, *ids = ord + nkeys, *cnt2 = ord + 2 * (size_t)nkeys, *ids2 = ord + 3 * (size_t)nkeys;
    hipLaunchKernelGGL(msm_counts_kernel, dim3((nkeys + T - 1) / T), dim3(T), 0, ctx->stream, cnt, ids, start, stop, nkeys, c);
    size_t tb = 0;
    XLP_HIPV(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, nul, nul, nul, nul, (int)nkeys, 0, 9, ctx->stream));
    const u32* order = ids;
    if (tb <= temp_bytes) {                                           // (the pair sort's temporary storage serves; always so beyond toy sizes)
        XLP_HIP(hipcub::DeviceRadixSort::SortPairs(temp, tb, cnt, cnt2, ids, ids2, (int)nkeys, 0, 9, ctx->stream));
        order = ids2;
    }
    hipLaunchKernelGGL(msm_buckets_kernel, dim3((nkeys + 63) / 64), dim3(64), 0, ctx->stream, buckets, start, stop, vals2, (const Aff*)bases, nkeys, c, heavy, hb, hs, order);
    hipLaunchKernelGGL(msm_heavy_kernel, dim3((unsigned)(nseg_max < 4096 ? nseg_max : 4096)), dim3(256), 0, ctx->stream, part, start, stop, vals2, (const Aff*)bases, heavy, hs);
    hipLaunchKernelGGL(msm_heavy_fold_kernel, dim3((unsigned)(nheavy_max < 1024 ? nheavy_max : 1024)), dim3(256), 0, ctx->stream, buckets, part, heavy, hb);
    hipLaunchKernelGGL(msm_groups_kernel, dim3((nwin * gpw + 63) / 64), dim3(64), 0, ctx->stream, groups, buckets, nwin, c);
    hipLaunchKernelGGL(msm_window_sums_kernel, dim3(nwin), dim3(256), 0, ctx->stream, wsum, groups, nwin, gpw);
    XLP_HIP(hipGetLastError());
    hostfq::J hw[128];                                                // (nwin <= 128: c >= 2)
    u64 res[8];
    XLP_HIP(hipMemcpyAsync(hw, wsum, (size_t)nwin * sizeof(Jac), hipMemcpyDeviceToHost, ctx->stream));
    XLP_HIP(hipStreamSynchronize(ctx->stream));
    hostfq::finish(res, hw, nwin, c);
    XLP_HIP(hipMemcpyAsync(out, res, sizeof res, hipMemcpyHostToDevice, ctx->stream));
    XLP_HIP(hipStreamSynchronize(ctx->stream));                       // (res lives on this stack)
    return 0;
}

u32 msm_bn254_window_bits(u64 n) { return msm_window_bits(n); }

// out (device, lanes x 12 words): the lanes' sums in Jacobian coordinates (kept so that the work is); point: one affine point (device, 8 words)
int dev_msm_resident(Ctx* ctx, u64* out, const u64* point, u32 lanes, u32 iters)
{
    if (!lanes) return 0;
    hipLaunchKernelGGL(g1::msm_resident_kernel, dim3((lanes + 63) / 64), dim3(64), 0, ctx->stream, (g1::Jac*)out, (const g1::Aff*)point, lanes, iters);
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
