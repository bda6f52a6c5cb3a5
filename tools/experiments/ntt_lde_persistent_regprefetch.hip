// EXPERIMENT (not built): persistent + register-prefetch variants of the LDE kernels.
// Result on MI355X (round 1): lde_top_dif 619 us vs 564 us for the one-tile-per-workgroup kernel --
// two waves/SIMD at 256 VGPRs lose more VALU issue than the overlapped loads win (pure-ALU time
// 451 us vs 336 us).  Kept for the record; see DESIGN.md 4.1.
// Persistent, register-prefetching variants of the three extendPol kernels (ntt_lde.hip) for the
// large sizes (BT >= 10, i.e. N >= 2^22).
//
// Why: a tile fills the CU's LDS (136-152 KiB), so only ONE workgroup is resident per CU and, in the
// one-tile-per-workgroup kernels, the HBM loads of a tile's first round are not overlapped with
// anything (measured 190 us of lde_top_dif's 560 us, profiles/r1/ablation_lde_kernels.txt).
// Here one 512-thread workgroup per CU walks over its tiles and, right after the first round has
// consumed its inputs, issues the NEXT tile's global loads into a second register set; they land
// while the remaining rounds of the current tile compute.  512 threads x 256 VGPRs: two waves per
// SIMD saturate the half-rate integer VALU as well as four do (measured), and leave room for the
// 64-VGPR prefetch buffer.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "ntt_common.hpp"
#include "ntt_lde_common.hpp"

namespace xlp {

#ifndef XLP_PABL
#define XLP_PABL 0    // ablation: 1 = no global stores, 2 = no prefetch loads (registers keep stale data), 3 = both
#endif
constexpr int PT = 512;   // threads per persistent workgroup

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), i.e. it would
// wait for the next tile's prefetch loads that were issued on purpose just before it.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// tile index of this workgroup's i-th tile: the workgroups co-resident on an XCD (bid % 8 equal)
// get consecutive tiles, and so do the 256 workgroups of one sweep.
__device__ __forceinline__ u32 sweep_tile(u32 i)
{
    const u32 bid = blockIdx.x, g = gridDim.x;   // g % 8 == 0
    return i * g + (bid & 7) * (g >> 3) + (bid >> 3);
}

// One radix-2^D round, PT threads, IT = NG / PT groups per thread (fully unrolled), generic sources.
//  SRC_REGS: inputs come from the prefetch registers nx[i][j] instead of load()
//  hook():   called after ALL twiddle loads of the round have been issued and before anything else.
//            vmcnt retires in order, so the next tile's prefetch must be issued AFTER the last global
//            load whose data this tile still needs soon -- i.e. here -- or the twiddles would queue
//            behind 128 KiB of prefetch traffic.
struct NoHook { __device__ __forceinline__ void operator()() const {} };

template <int B, int V, int TP, int D, bool DIT, bool INV, bool HAS_PRE, bool HAS_POST, bool POSTW, bool SRC_REGS, bool SRC_CANON = false, int NXI, int NXJ,
          class Load, class Store, class Pre, class Post, class Hook = NoHook>
__device__ __forceinline__ void round_p(const u64* __restrict__ tw, u64 (&nx)[NXI][NXJ], Load load, Store store, Pre pre, Post post, Hook hook = Hook{})
{
    constexpr int RAD = 1 << D;
    constexpr u32 VB = V == 8 ? 3 : 2;
    constexpr u32 NG = (1u << (B - D)) * V;
    static_assert(NG % PT == 0, "groups must divide over the workgroup");
    constexpr int IT = NG / PT;
    constexpr u32 TSHIFT = B - TP - D;
    constexpr u32 RS = TP >= 4 ? (1u << TP) + ((1u << TP) >> 4) : (1u << TP);
    u64 twr[IT][RAD];
#pragma unroll
    for (int i = 0; i < IT; i++) {
        const u32 g = threadIdx.x + i * PT;
        const u32 rest = g >> VB;
        const u32 low = rest & ((1u << TP) - 1);
        const u32 high = rest >> TP;
        const u32 base = (high << (TP + D)) | low;
        if constexpr (POSTW) {
#pragma unroll
            for (int j = 0; j < RAD; j++) twr[i][j] = tw[base + ((u32)j << TP)];
        } else if constexpr (TP > 0) {
#pragma unroll
            for (int j = 1; j < RAD; j++) twr[i][j] = tw[((u32)j * low) << TSHIFT];
        }
    }
    hook();
#pragma unroll
    for (int i = 0; i < IT; i++) {
        const u32 g = threadIdx.x + i * PT;
        const u32 v = g & (V - 1);
        const u32 rest = g >> VB;
        const u32 low = rest & ((1u << TP) - 1);
        const u32 high = rest >> TP;
        const u32 base = (high << (TP + D)) | low;
        const u32 r0 = lds_row(base);
        u64 x[RAD];
#pragma unroll
        for (int j = 0; j < RAD; j++) {
            if constexpr (SRC_REGS) x[j] = SRC_CANON ? gl::canon(nx[i][j]) : nx[i][j];
            else x[j] = load(base + ((u32)j << TP), v, r0 + (u32)j * RS);
        }
        if constexpr (HAS_PRE) {
            u64 f[RAD];
#pragma unroll
            for (int j = 0; j < RAD; j++) f[j] = pre(base + ((u32)j << TP));
#pragma unroll
            for (int j = 0; j < RAD; j++) x[j] = gl::mulc(x[j], f[j]);
        }
        if constexpr (POSTW) {
#pragma unroll
            for (int j = 0; j < RAD; j++) x[j] = gl::mulc(x[j], twr[i][j]);
        } else if constexpr (DIT && TP > 0) {
#pragma unroll
            for (int j = 1; j < RAD; j++) x[j] = gl::mulc(x[j], twr[i][j]);
        }
        dft_regs<D>(x);
        u64 y[RAD];
#pragma unroll
        for (int k = 0; k < RAD; k++) y[k] = x[brev<D>(INV ? (RAD - k) % RAD : k)];
        if constexpr (!DIT && TP > 0) {
#pragma unroll
            for (int k = 1; k < RAD; k++) y[k] = gl::mulc(y[k], twr[i][k]);
        }
        if constexpr (HAS_POST) {
            u64 f[RAD];
#pragma unroll
            for (int k = 0; k < RAD; k++) f[k] = post(base + ((u32)k << TP));
#pragma unroll
            for (int k = 0; k < RAD; k++) y[k] = gl::mulc(y[k], f[k]);
        }
#pragma unroll
        for (int k = 0; k < RAD; k++) store(base + ((u32)k << TP), v, y[k], r0 + (u32)k * RS);
    }
}

// issue the global loads of one tile's first round into nx (no wait): addr(tpos, v) -> element pointer
template <int B, int V, int TP, int D, int NXI, int NXJ, class Addr>
__device__ __forceinline__ void prefetch_p(u64 (&nx)[NXI][NXJ], Addr addr)
{
    constexpr u32 VB = V == 8 ? 3 : 2;
    constexpr u32 NG = (1u << (B - D)) * V;
    constexpr int IT = NG / PT;
    static_assert(IT == NXI && (1 << D) == NXJ, "prefetch buffer shape");
#pragma unroll
    for (int i = 0; i < IT; i++) {
        const u32 g = threadIdx.x + i * PT;
        const u32 v = g & (V - 1);
        const u32 rest = g >> VB;
        const u32 low = rest & ((1u << TP) - 1);
        const u32 high = rest >> TP;
        const u32 base = (high << (TP + D)) | low;
#pragma unroll
        for (int j = 0; j < (1 << D); j++) {
#if XLP_PABL & 2
            nx[i][j] = (u64)(size_t)addr(base + ((u32)j << TP), v);
#else
            nx[i][j] = *addr(base + ((u32)j << TP), v);
#endif
        }
    }
}

// ------------------------------------------------------------------------ K1 (persistent)
template <int BT>
__global__ __launch_bounds__(PT) void lde_top_dif_p(const LdeParams p, u32 ntiles)
{
    constexpr int V = 8, m = 3, dtop = BT - 8;
    static_assert(BT >= 10 && BT <= 12, "persistent top kernels: 3 rounds");
    constexpr int IT0 = ((1u << (BT - dtop)) * V) / PT, R0 = 1 << dtop;
    extern __shared__ u64 smem[];
    u64* sm = smem;
    u64* cross = smem + ((1u << BT) + (1u << BT) / 16 + 1) * V;
    const u32 ncc = (p.wc + V - 1) / V;
    u64 nx[IT0][R0];
    u64 dummy[1][1];
    u32 tile = sweep_tile(0);
    auto addr_of = [&](u32 t) {
        const u32 cc = t % ncc, lo = t / ncc;
        const u32 c = cc * V;
        return [=, &p](u32 tpos, u32 v) -> const u64* {
            const u32 col = c + v < p.wc ? c + v : p.wc - 1;
            const u64 row = ((u64)tpos << 12) | lo;
            return p.in + row * p.ld_in + p.col0 + col;
        };
    };
    if (tile < ntiles) prefetch_p<BT, V, 8, dtop>(nx, addr_of(tile));
    for (u32 it = 0; tile < ntiles; it++) {
        const u32 cc = tile % ncc, lo = tile / ncc;
        const u32 c = cc * V;
        for (u32 k = threadIdx.x; k < (1u << BT); k += PT) cross[k] = gl::canon(two_level12(p.cri_lo, p.cri_hi, (u64)k * lo));
        auto lload = [&](u32, u32 v, u32 lrow) -> u64 { return sm[lrow * V + v]; };
        auto lstore = [&](u32, u32 v, u64 val, u32 lrow) { sm[lrow * V + v] = val; };
        auto gstore = [&](u32 tpos, u32 v, u64 val, u32) {
            const u64 row = ((u64)tpos << 12) | lo;
#if XLP_PABL & 1
            if (val == 0x1234567ULL) p.A[row * p.wc + c + v] = val;
#else
            if (c + v < p.wc) p.A[row * p.wc + c + v] = val;
#endif
        };
        auto crossf = [&](u32 tpos) -> u64 { return cross[tile_natural_c<BT>(tpos)]; };
        // round 0 from the prefetched registers
        round_p<BT, V, 8, dtop, false, true, false, false, false, true, true>(p.twi_bt, nx, lload, lstore, NoFactor{}, NoFactor{});
        const u32 next = sweep_tile(it + 1);
        lds_barrier();
        // next tile's inputs start streaming once this round's twiddle loads are in flight
        // unconditional (a branch would make hipcc merge the wait counters conservatively): past the end, re-read this tile
        auto pf = [&]() { prefetch_p<BT, V, 8, dtop>(nx, addr_of(next < ntiles ? next : tile)); };
        round_p<BT, V, 4, 4, false, true, false, false, false, false>(p.twi_bt, dummy, lload, lstore, NoFactor{}, NoFactor{}, pf);
        lds_barrier();
        round_p<BT, V, 0, 4, false, true, false, true, false, false>(p.twi_bt, dummy, lload, gstore, NoFactor{}, crossf);
        lds_barrier();   // LDS tile and cross vector are reused by the next tile
        tile = next;
    }
    (void)m;
}

// ------------------------------------------------------------------------ K3 (persistent)
template <int BT>
__global__ __launch_bounds__(PT) void lde_top_dit_p(const LdeParams p, u32 ntiles)
{
    constexpr int V = 8, dtop = BT - 8;
    static_assert(BT >= 10 && BT <= 12, "persistent top kernels: 3 rounds");
    constexpr int IT0 = ((1u << (BT - 4)) * V) / PT;
    extern __shared__ u64 smem[];
    u64* sm = smem;
    u64* cross = smem + ((1u << BT) + (1u << BT) / 16 + 1) * V;
    const u32 ncc = (p.wc + V - 1) / V;
    const u32 ncos = 1u << p.e;
    const u64 n = 1ull << p.L;
    u64 nx[IT0][16];
    u64 dummy[1][1];
    u32 tile = sweep_tile(0);
    auto addr_of = [&](u32 t) {
        const u32 r = t % ncos, cc = (t / ncos) % ncc, lo = t / (ncos * ncc);
        const u32 c = cc * V;
        const u64* Br = p.B + (u64)r * n * p.wc;
        return [=, &p](u32 tpos, u32 v) -> const u64* {
            const u32 col = c + v < p.wc ? c + v : p.wc - 1;
            const u64 row = ((u64)tpos << 12) | lo;
            return Br + row * p.wc + col;
        };
    };
    if (tile < ntiles) prefetch_p<BT, V, 0, 4>(nx, addr_of(tile));
    for (u32 it = 0; tile < ntiles; it++) {
        const u32 r = tile % ncos, cc = (tile / ncos) % ncc, lo = tile / (ncos * ncc);
        const u32 c = cc * V;
        const u64* __restrict__ ur = p.ur[r];
        for (u32 j = threadIdx.x; j < (1u << BT); j += PT) cross[j] = gl::mulc(two_level12(p.crf_lo, p.crf_hi, (u64)j * lo), ur[j]);
        lds_barrier();
        auto crossf = [&](u32 tpos) -> u64 { return cross[tile_natural_c<BT>(tpos)]; };
        auto lload = [&](u32, u32 v, u32 lrow) -> u64 { return sm[lrow * V + v]; };
        auto lstore = [&](u32, u32 v, u64 val, u32 lrow) { sm[lrow * V + v] = val; };
        auto gstore = [&](u32 tpos, u32 v, u64 val, u32) {
            const u64 row = (((((u64)tpos << 12) | lo)) << p.e) | r;
#if XLP_PABL & 1
            if (val == 0x1234567ULL) p.out[row * p.ld_out + p.col0 + c + v] = val;
#else
            if (c + v < p.wc) p.out[row * p.ld_out + p.col0 + c + v] = val;
#endif
        };
        round_p<BT, V, 0, 4, true, false, true, false, false, true>(p.twf_bt, nx, lload, lstore, crossf, NoFactor{});
        const u32 next = sweep_tile(it + 1);
        lds_barrier();
        auto pf = [&]() { prefetch_p<BT, V, 0, 4>(nx, addr_of(next < ntiles ? next : tile)); };
        round_p<BT, V, 4, 4, true, false, false, false, false, false>(p.twf_bt, dummy, lload, lstore, NoFactor{}, NoFactor{}, pf);
        lds_barrier();
        round_p<BT, V, 8, dtop, true, false, false, false, false, false>(p.twf_bt, dummy, lload, gstore, NoFactor{}, NoFactor{});
        lds_barrier();
        tile = next;
    }
}

// ------------------------------------------------------------------------ K2 (persistent)
__global__ __launch_bounds__(PT) void lde_mid_p(const LdeParams p, u32 ntiles)
{
    constexpr int V = 4, B = 12, NGT = 1024 / PT;
    extern __shared__ u64 sm[];
    const u32 ncc = (p.wc + V - 1) / V;
    const u64 n = 1ull << p.L;
    u64 nx[NGT][16];
    u64 dummy[1][1];
    u32 tile = sweep_tile(0);
    auto addr_of = [&](u32 t) {
        const u32 cc = t % ncc, hi = t / ncc;
        const u32 c = cc * V;
        return [=, &p](u32 tpos, u32 v) -> const u64* {
            const u32 col = c + v < p.wc ? c + v : p.wc - 1;
            const u64 row = ((u64)hi << 12) | tpos;
            return p.A + row * p.wc + col;
        };
    };
    if (tile < ntiles) prefetch_p<B, V, 8, 4>(nx, addr_of(tile));
    const u32 v = threadIdx.x & 3;
    for (u32 it = 0; tile < ntiles; it++) {
        const u32 cc = tile % ncc, hi = tile / ncc;
        const u32 c = cc * V;
        auto lload = [&](u32, u32 vv, u32 lrow) -> u64 { return sm[lrow * V + vv]; };
        auto lstore = [&](u32, u32 vv, u64 val, u32 lrow) { sm[lrow * V + vv] = val; };
        round_p<B, V, 8, 4, false, true, false, false, false, true>(p.twi_12, nx, lload, lstore, NoFactor{}, NoFactor{});
        const u32 next = sweep_tile(it + 1);
        lds_barrier();
        auto pf = [&]() { prefetch_p<B, V, 8, 4>(nx, addr_of(next < ntiles ? next : tile)); };
        round_p<B, V, 4, 4, false, true, false, false, false, false>(p.twi_12, dummy, lload, lstore, NoFactor{}, NoFactor{}, pf);
        lds_barrier();
        u32 base_[NGT];
        u64 X[NGT][16];
#pragma unroll
        for (int i = 0; i < NGT; i++) {
            base_[i] = ((threadIdx.x + i * PT) >> 2) << 4;
            const u32 base = base_[i];
            u64 x[16];
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = sm[(lds_row(base) + j) * V + v];
            dft_regs<4>(x);
#pragma unroll
            for (int k = 0; k < 16; k++) X[i][k] = x[brev<4>((16 - k) % 16)];
        }
        for (u32 r = 0; r < (1u << p.e); r++) {
            const u64* __restrict__ hk = p.hk[r];
            u64* __restrict__ Br = p.B + (u64)r * n * p.wc;
            lds_barrier();
#pragma unroll
            for (int i = 0; i < NGT; i++) {
                const u32 base = base_[i];
                u64 x[16];
                x[0] = X[i][0];
#pragma unroll
                for (int j = 1; j < 16; j++) x[j] = gl::mulc(X[i][j], hk[j]);
                dft_regs<4>(x);
#pragma unroll
                for (int k = 0; k < 16; k++) sm[(lds_row(base) + k) * V + v] = x[brev<4>(k)];
            }
            lds_barrier();
            round_p<B, V, 4, 4, true, false, false, false, true, false>(p.tw2[r], dummy, lload, lstore, NoFactor{}, NoFactor{});
            lds_barrier();
            auto gstore = [&](u32 tpos, u32 vv, u64 val, u32) {
                const u64 row = ((u64)hi << 12) | tpos;
                if (c + vv < p.wc) Br[row * p.wc + c + vv] = val;
            };
            round_p<B, V, 8, 4, true, false, false, false, false, false>(p.twf_12, dummy, lload, gstore, NoFactor{}, NoFactor{});
        }
        lds_barrier();
        tile = next;
    }
}

// ----------------------------------------------------------------- launchers
template <int BT>
static int launch_top_p(Ctx* ctx, const LdeParams& p, bool dit, u32 ncu)
{
    constexpr u32 V = 8;
    const u32 ncc = (p.wc + V - 1) / V;
    const size_t lds = ((size_t)((1u << BT) + (1u << BT) / 16 + 1) * V + (1u << BT)) * sizeof(u64);
    static bool attr_done = false;
    if (!attr_done) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_top_dif_p<BT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        XLP_HIP(hipFuncSetAttribute((const void*)lde_top_dit_p<BT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const u64 ntiles = dit ? (((u64)ncc << 12) << p.e) : ((u64)ncc << 12);
    const u32 grid = ntiles < ncu ? (u32)((ntiles + 7) / 8 * 8) : ncu;
    if (!dit) hipLaunchKernelGGL(lde_top_dif_p<BT>, dim3(grid), dim3(PT), lds, ctx->stream, p, (u32)ntiles);
    else hipLaunchKernelGGL(lde_top_dit_p<BT>, dim3(grid), dim3(PT), lds, ctx->stream, p, (u32)ntiles);
    XLP_HIP(hipGetLastError());
    return 0;
}

bool lde_persistent_supported(u32 BT) { return BT >= 10 && BT <= 11; }

int launch_lde_top_p(Ctx* ctx, const LdeParams& p, u32 BT, bool dit, u32 ncu)
{
    switch (BT) {
        case 10: return launch_top_p<10>(ctx, p, dit, ncu);
        case 11: return launch_top_p<11>(ctx, p, dit, ncu);
        default: return xlp_fail("lde persistent path: unsupported size");
    }
}

int launch_lde_mid_p(Ctx* ctx, const LdeParams& p, u32 BT, u32 ncu)
{
    static bool attr = false;
    if (!attr) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_mid_p, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const u32 ncc = (p.wc + 3) / 4;
    const u64 ntiles = (u64)ncc << BT;
    const u32 grid = ntiles < ncu ? (u32)((ntiles + 7) / 8 * 8) : ncu;
    const size_t lds = (size_t)(4096 + 256 + 1) * 4 * sizeof(u64);
    hipLaunchKernelGGL(lde_mid_p, dim3(grid), dim3(PT), lds, ctx->stream, p, (u32)ntiles);
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
