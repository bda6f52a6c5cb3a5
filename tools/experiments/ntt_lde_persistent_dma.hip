// EXPERIMENT (not built by default; add to SRCS with -DXLP_LDE_PERSISTENT_EXPERIMENT).  Measured on MI355X, round 2
// (profiles/r2/lde_persistent_dma_ab.json): lde_top_dif 0.465 -> 0.488 ms, lde_mid 1.19 -> 1.67 ms, lde_top_dit 0.95 -> 1.13 ms at 2^23 x 8:
// slower.  VMEM operations retire in issue order, so the first twiddle / cross-table load of the next tile waits for the
// previous tile's stores to drain (a fresh workgroup has fresh counters), the 256 workgroups stay phase-locked (bursts instead
// of a steady mix), and lde_mid spills inside the tile loop.
// Persistent forms of the three extendPol kernels (ntt_lde.hip) for N >= 2^22 (BT >= 10), where one tile fills
// the CU's LDS and only ONE workgroup is resident per CU: in the one-tile-per-workgroup kernels the HBM loads
// of a tile's first round overlap with nothing (profiles/r1/ablation_lde_kernels.txt: 0.6 ms of 2.7 ms).
//
// Here one 1024-thread workgroup per CU walks over its tiles.  In the LAST round of a tile every thread first
// reads its butterfly inputs from LDS; after that barrier the tile is dead, and the workgroup fills it with the
// NEXT tile straight from HBM with LDS-direct loads (global_load_lds_dwordx4: no VGPR destination, so the
// 128-VGPR budget of 4 waves/SIMD is untouched -- the register-prefetch form of round 1 lost more issue slots
// than it won).  The loads land while the last round's arithmetic and global stores run; the next tile's first
// round then reads LDS instead of HBM.  The LDS image is the padded tile itself: one wave instruction writes
// 1 KiB = 16 rows x 64 B (V = 8) exactly between two pad rows; the per-lane SOURCE address carries the row
// stride.  Tiles whose column chunk is ragged (fewer than V valid columns) are filled with ordinary loads.
//
// Tile order: sweep i gives workgroup b tile i*G + (b%8)*(G/8) + b/8, so the 32 workgroups that share an XCD
// (and its L2) work on 32 consecutive tiles = adjacent rows / column chunks / cosets.
//
// Replaces (same arithmetic, same tables, same scratch layout) lde_top_dif / lde_mid / lde_top_dit of
// ntt_lde.hip, i.e. NTT_Goldilocks::extendPol as called at starks.cpp:53,134,215.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "ntt_common.hpp"
#include "ntt_lde_common.hpp"

namespace xlp {

constexpr int PT = 1024;   // threads per persistent workgroup (16 waves, 4 per SIMD, <= 128 VGPRs)

__device__ __forceinline__ u32 sweep_tile(u32 i)
{
    const u32 bid = blockIdx.x, g = gridDim.x;   // g % 8 == 0
    return i * g + (bid & 7) * (g >> 3) + (bid >> 3);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// LDS-direct fill of a padded tile of NROWS rows x (V x 8) bytes.  rowptr(tpos) -> first of the V columns of row tpos.
//   V = 8: instruction k covers rows 16k..16k+15 (4 lanes x 16 B per row)       -> LDS word (17 k) * 8
//   V = 4: instruction k covers rows 16k..16k+15 with lanes 0..31 (2 per row)  -> LDS word (17 k) * 4
template <int NROWS, int V, class RowPtr>
__device__ __forceinline__ void tile_fill_dma(u64* sm, RowPtr rowptr)
{
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr u32 LPR = V / 2;                     // lanes per row
    constexpr u32 NI = NROWS / 16;
    if (V == 8 || lane < 32) {
#pragma unroll 1
        for (u32 k = wave; k < NI; k += PT / 64) {
            const u64* g = rowptr(16 * k + lane / LPR) + (lane % LPR) * 2;
            // inline asm, so that hipcc does not track the load: its own bookkeeping would drain it (and the stores issued
            // behind it) with vmcnt(0) at the loop-top barrier and at the first LDS read of the tile.  wait_tile() counts instead.
            const u32 lds_off = __builtin_amdgcn_readfirstlane((u32)(k * 17 * V * 8));
            u32 m0_saved;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_saved) : "v"(g), "s"(lds_off) : "memory");
        }
    }
}

// the same fill with ordinary loads (ragged column chunks): elem(tpos, v) -> value (column clamped by the caller)
template <int NROWS, int V, class Elem>
__device__ __forceinline__ void tile_fill_plain(u64* sm, Elem elem)
{
    for (u32 g = threadIdx.x; g < NROWS * V; g += PT) {
        const u32 tpos = g / V, v = g % V;
        sm[lds_row(tpos) * V + v] = elem(tpos, v);
    }
}

// The filled tile has landed: every wave waits for its own LDS-direct loads, then the workgroup meets.  VMEM operations retire
// in issue order on gfx9, and the only ones a wave issues behind its LDS-direct loads are the 16 global stores of the last
// round (full tiles: every wave stores): vmcnt(16) waits for the loads and leaves the stores in flight.
__device__ __forceinline__ void wait_tile(bool stores_behind)
{
    if (stores_behind) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct NoHook { __device__ __forceinline__ void operator()() const {} };

// One radix-2^D round, PT threads, IT groups per thread.  Without a hook the groups run one after the other (as in
// fast_round of ntt_lde.hip).  With a hook: (1) every twiddle and data load of the thread's IT groups, (2) hook() --
// executed by ALL threads; the last round's hook is "barrier, then start filling the dead tile with the next one" --
// (3) arithmetic, (4) stores.  The loaded values are pinned in front of the hook so that no ordinary global load is
// still pending when the LDS-direct loads are issued (hipcc would drain them with vmcnt(0) at the first use of a
// pending load's result).  LDS reads behind the hook (the cross-twiddle vector) do not wait for the LDS-direct loads.
template <int B, int V, int TP, int D, bool DIT, bool INV, bool HAS_PRE, bool HAS_POST, bool POSTW, bool CANON_IN,
          class Load, class Store, class Pre, class Post, class Hook>
__device__ __forceinline__ void round_p(const u32 tid, const u64* __restrict__ tw, Load load, Store store, Pre pre, Post post, Hook hook)
{
    constexpr int RAD = 1 << D;
    constexpr u32 VB = V == 8 ? 3 : 2;
    constexpr u32 NG = (1u << (B - D)) * V;
    constexpr int IT = (NG + PT - 1) / PT;
    constexpr bool FULL = (NG % PT) == 0;
    constexpr bool HOOKED = !std::is_same<Hook, NoHook>::value;
    constexpr int NB = HOOKED ? IT : 1;            // groups whose loads are in flight together
    constexpr u32 TSHIFT = B - TP - D;
    constexpr u32 RS = TP >= 4 ? (1u << TP) + ((1u << TP) >> 4) : (1u << TP);
    static_assert(TP == 0 || TP >= 4, "digit offsets are multiples of 4");
#pragma unroll 1
    for (int i0 = 0; i0 < IT; i0 += NB) {
        u64 twr[NB][RAD];
        u64 x[NB][RAD];
        u32 base_[NB], v_[NB], r0_[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const u32 g = tid + (i0 + i) * PT;
            const bool act = FULL || g < NG;
            const u32 gg = act ? g : 0;
            v_[i] = gg & (V - 1);
            const u32 rest = gg >> VB;
            const u32 low = rest & ((1u << TP) - 1);
            const u32 high = rest >> TP;
            base_[i] = (high << (TP + D)) | low;
            r0_[i] = lds_row(base_[i]);
            if constexpr (POSTW) {
#pragma unroll
                for (int j = 0; j < RAD; j++) twr[i][j] = tw[base_[i] + ((u32)j << TP)];
            } else if constexpr (TP > 0) {
#pragma unroll
                for (int j = 1; j < RAD; j++) twr[i][j] = tw[((u32)j * low) << TSHIFT];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
#pragma unroll
            for (int j = 0; j < RAD; j++) x[i][j] = load(base_[i] + ((u32)j << TP), v_[i], r0_[i] + (u32)j * RS);
        }
        if constexpr (HOOKED) {
#pragma unroll
            for (int i = 0; i < NB; i++) {
#pragma unroll
                for (int j = (POSTW ? 0 : 1); j < RAD; j++)
                    if constexpr (POSTW || TP > 0) asm volatile("" : "+v"(twr[i][j]));
#pragma unroll
                for (int j = 0; j < RAD; j++) asm volatile("" : "+v"(x[i][j]));
            }
            hook();
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const u32 g = tid + (i0 + i) * PT;
            const bool act = FULL || g < NG;
            const u32 base = base_[i];
            if constexpr (CANON_IN) {
#pragma unroll
                for (int j = 0; j < RAD; j++) x[i][j] = gl::canon(x[i][j]);
            }
            if constexpr (HAS_PRE) {
                u64 f[RAD];
#pragma unroll
                for (int j = 0; j < RAD; j++) f[j] = pre(base + ((u32)j << TP));
#pragma unroll
                for (int j = 0; j < RAD; j++) x[i][j] = gl::mulc(x[i][j], f[j]);
            }
            if constexpr (POSTW) {
#pragma unroll
                for (int j = 0; j < RAD; j++) x[i][j] = gl::mulc(x[i][j], twr[i][j]);
            } else if constexpr (DIT && TP > 0) {
#pragma unroll
                for (int j = 1; j < RAD; j++) x[i][j] = gl::mulc(x[i][j], twr[i][j]);
            }
            dft_regs<D>(x[i]);
            u64 y[RAD];
#pragma unroll
            for (int k = 0; k < RAD; k++) y[k] = x[i][brev<D>(INV ? (RAD - k) % RAD : k)];
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
            if (act) {
#pragma unroll
                for (int k = 0; k < RAD; k++) store(base + ((u32)k << TP), v_[i], y[k], r0_[i] + (u32)k * RS);
            }
        }
    }
}

// ------------------------------------------------------------------------ K1
template <int BT, bool INV>
__global__ __launch_bounds__(PT) void lde_top_dif_p(const LdeParams p, const u32 ntiles)
{
    constexpr int V = 8, dtop = BT - 8;
    static_assert(BT >= 9 && BT <= 12, "three rounds: dtop,4,4");
    extern __shared__ u64 sm[];
    u64* cross = sm + ((1u << BT) + (1u << BT) / 16 + 1) * V;
    const u32 ncc = (p.wc + V - 1) / V;
    auto lload = [&](u32, u32 v, u32 lrow) -> u64 { return sm[lrow * V + v]; };
    auto lstore = [&](u32, u32 v, u64 val, u32 lrow) { sm[lrow * V + v] = val; };
    // start the fill of tile t (asynchronous when its 8 columns are all valid)
    auto fill = [&](u32 t) {
        const u32 cc = t % ncc, lo = t / ncc, c = cc * V;
        if (c + V <= p.wc) {
            tile_fill_dma<(1 << BT), V>(sm, [&](u32 tpos) { return p.in + (((u64)tpos << 12) | lo) * p.ld_in + p.col0 + c; });
        } else {
            tile_fill_plain<(1 << BT), V>(sm, [&](u32 tpos, u32 v) {
                const u32 col = c + v < p.wc ? c + v : p.wc - 1;
                return p.in[(((u64)tpos << 12) | lo) * p.ld_in + p.col0 + col];
            });
        }
    };
    u32 it = 0, tile = sweep_tile(0);
    if (tile >= ntiles) return;
    fill(tile);
    bool counted = false;    // 16 store instructions were issued behind this tile's LDS-direct loads
    for (;;) {
        const u32 cc = tile % ncc, lo = tile / ncc, c = cc * V;
        const u32 next = sweep_tile(++it);
        wait_tile(counted);   // the tile has landed; nobody reads the previous cross vector any more
        u32 tid = threadIdx.x;
        asm volatile("" : "+v"(tid));   // per-thread addresses are recomputed per tile instead of living (and spilling) across the tile loop
        for (u32 k = threadIdx.x; k < (1u << BT); k += PT) cross[k] = gl::canon(two_level12(p.cri_lo, p.cri_hi, (u64)k * lo));
        round_p<BT, V, 8, dtop, false, INV, false, false, false, true>(tid, p.twi_bt, lload, lstore, NoFactor{}, NoFactor{}, NoHook{});
        __syncthreads();
        round_p<BT, V, 4, 4, false, INV, false, false, false, false>(tid, p.twi_bt, lload, lstore, NoFactor{}, NoFactor{}, NoHook{});
        __syncthreads();
        auto gstore = [&](u32 tpos, u32 v, u64 val, u32) {
            const u64 row = ((u64)tpos << 12) | lo;
            if (c + v < p.wc) p.A[row * p.wc + c + v] = val;
        };
        auto crossf = [&](u32 tpos) -> u64 { return cross[tile_natural_c<BT>(tpos)]; };
        auto hook = [&]() {
            __syncthreads();                      // every thread holds its inputs: the tile is dead
            if (next < ntiles) fill(next);
        };
        round_p<BT, V, 0, 4, false, INV, false, true, false, false>(tid, p.twi_bt, lload, gstore, NoFactor{}, crossf, hook);
        if (next >= ntiles) break;
        counted = c + V <= p.wc;     // every wave issued its 16 stores behind the prefetch
        tile = next;
    }
}

// ------------------------------------------------------------------------ K3
template <int BT>
__global__ __launch_bounds__(PT) void lde_top_dit_p(const LdeParams p, const u32 ntiles)
{
    constexpr int V = 8, dtop = BT - 8;
    extern __shared__ u64 sm[];
    u64* cross = sm + ((1u << BT) + (1u << BT) / 16 + 1) * V;
    const u32 ncc = (p.wc + V - 1) / V;
    const u32 ncos = 1u << p.e;
    const u64 n = 1ull << p.L;
    auto lload = [&](u32, u32 v, u32 lrow) -> u64 { return sm[lrow * V + v]; };
    auto lstore = [&](u32, u32 v, u64 val, u32 lrow) { sm[lrow * V + v] = val; };
    auto fill = [&](u32 t) {
        const u32 r = t % ncos, cc = (t / ncos) % ncc, lo = t / (ncos * ncc), c = cc * V;
        const u64* __restrict__ Br = p.B + (u64)r * n * p.wc;
        if (c + V <= p.wc) {
            tile_fill_dma<(1 << BT), V>(sm, [&](u32 tpos) { return Br + (((u64)tpos << 12) | lo) * p.wc + c; });
        } else {
            tile_fill_plain<(1 << BT), V>(sm, [&](u32 tpos, u32 v) {
                const u32 col = c + v < p.wc ? c + v : p.wc - 1;
                return Br[(((u64)tpos << 12) | lo) * p.wc + col];
            });
        }
    };
    u32 it = 0, tile = sweep_tile(0);
    if (tile >= ntiles) return;
    fill(tile);
    bool counted = false;    // 16 store instructions were issued behind this tile's LDS-direct loads
    for (;;) {
        const u32 r = tile % ncos, cc = (tile / ncos) % ncc, lo = tile / (ncos * ncc), c = cc * V;
        const u32 next = sweep_tile(++it);
        const u64* __restrict__ ur = p.ur[r];
        wait_tile(counted);
        u32 tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        for (u32 j = threadIdx.x; j < (1u << BT); j += PT)
            cross[j] = gl::mulc(two_level12(p.crf_lo, p.crf_hi, (u64)j * lo), ur[j]);
        __syncthreads();
        auto crossf = [&](u32 tpos) -> u64 { return cross[tile_natural_c<BT>(tpos)]; };
        round_p<BT, V, 0, 4, true, false, true, false, false, false>(tid, p.twf_bt, lload, lstore, crossf, NoFactor{}, NoHook{});
        __syncthreads();
        round_p<BT, V, 4, 4, true, false, false, false, false, false>(tid, p.twf_bt, lload, lstore, NoFactor{}, NoFactor{}, NoHook{});
        __syncthreads();
        auto gstore = [&](u32 tpos, u32 v, u64 val, u32) {
            const u64 row = (((((u64)tpos << 12) | lo)) << p.e) | r;
            if (c + v < p.wc) p.out[row * p.ld_out + p.col0 + c + v] = val;
        };
        auto hook = [&]() {
            __syncthreads();
            if (next < ntiles) fill(next);
        };
        round_p<BT, V, 8, dtop, true, false, false, false, false, false>(tid, p.twf_bt, lload, gstore, NoFactor{}, NoFactor{}, hook);
        if (next >= ntiles) break;
        counted = c + V <= p.wc;     // every wave issued its 16 stores behind the prefetch
        tile = next;
    }
}

// ------------------------------------------------------------------------ K2
__global__ __launch_bounds__(PT) void lde_mid_p(const LdeParams p, const u32 ntiles)
{
    constexpr int V = 4, B = 12;
    extern __shared__ u64 sm[];
    const u32 ncc = (p.wc + V - 1) / V;
    const u64 n = 1ull << p.L;
    auto lload = [&](u32, u32 v, u32 lrow) -> u64 { return sm[lrow * V + v]; };
    auto lstore = [&](u32, u32 v, u64 val, u32 lrow) { sm[lrow * V + v] = val; };
    auto fill = [&](u32 t) {
        const u32 cc = t % ncc, hi = t / ncc, c = cc * V;
        if (c + V <= p.wc) {
            tile_fill_dma<(1 << B), V>(sm, [&](u32 tpos) { return p.A + (((u64)hi << 12) | tpos) * p.wc + c; });
        } else {
            tile_fill_plain<(1 << B), V>(sm, [&](u32 tpos, u32 v) {
                const u32 col = c + v < p.wc ? c + v : p.wc - 1;
                return p.A[(((u64)hi << 12) | tpos) * p.wc + col];
            });
        }
    };
    u32 it = 0, tile = sweep_tile(0);
    if (tile >= ntiles) return;
    fill(tile);
    bool counted = false;    // 16 store instructions were issued behind this tile's LDS-direct loads
    const u32 ncos = 1u << p.e;
    for (;;) {
        const u32 cc = tile % ncc, hi = tile / ncc, c = cc * V;
        const u32 next = sweep_tile(++it);
        wait_tile(counted);
        u32 tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const u32 v = tid & 3;
        const u32 base = (tid >> 2) << 4;     // the thread's group in a TP = 0 round
        // ---- inverse NTT over the low 12 bits (DIF); last round keeps its outputs in VGPRs
        round_p<B, V, 8, 4, false, true, false, false, false, false>(tid, p.twi_12, lload, lstore, NoFactor{}, NoFactor{}, NoHook{});
        __syncthreads();
        round_p<B, V, 4, 4, false, true, false, false, false, false>(tid, p.twi_12, lload, lstore, NoFactor{}, NoFactor{}, NoHook{});
        __syncthreads();
        u64 X[16];
        {
            u64 x[16];
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = sm[(lds_row(base) + j) * V + v];
            dft_regs<4>(x);
#pragma unroll
            for (int k = 0; k < 16; k++) X[k] = x[brev<4>((16 - k) % 16)];
        }
        // ---- every coset: scale, forward NTT over the same 12 bits (DIT), -> B_r   (see lde_mid in ntt_lde.hip).
        // The last coset is peeled: the coefficients X die with its scaling, and its last round carries the prefetch.
        auto coset = [&](const u32 r, auto hook) {
            const u64* __restrict__ hk = p.hk[r];
            u64* __restrict__ Br = p.B + (u64)r * n * p.wc;
            __syncthreads();
            {
                u64 x[16];
                x[0] = X[0];
#pragma unroll
                for (int j = 1; j < 16; j++) x[j] = gl::mulc(X[j], hk[j]);
                dft_regs<4>(x);
#pragma unroll
                for (int k = 0; k < 16; k++) sm[(lds_row(base) + k) * V + v] = x[brev<4>(k)];
            }
            __syncthreads();
            round_p<B, V, 4, 4, true, false, false, false, true, false>(tid, p.tw2[r], lload, lstore, NoFactor{}, NoFactor{}, NoHook{});
            __syncthreads();
            auto gstore = [&](u32 tpos, u32 vv, u64 val, u32) {
                const u64 row = ((u64)hi << 12) | tpos;
                if (c + vv < p.wc) Br[row * p.wc + c + vv] = val;
            };
            round_p<B, V, 8, 4, true, false, false, false, false, false>(tid, p.twf_12, lload, gstore, NoFactor{}, NoFactor{}, hook);
        };
#pragma unroll 1
        for (u32 r = 0; r + 1 < ncos; r++) coset(r, NoHook{});
        coset(ncos - 1, [&]() {
            __syncthreads();
            if (next < ntiles) fill(next);
        });
        if (next >= ntiles) break;
        counted = c + V <= p.wc;     // every wave issued its 16 stores behind the prefetch
        tile = next;
    }
}

// ----------------------------------------------------------------- host side
// BT = 10 tiles leave room for two one-tile workgroups per CU, which overlap by themselves (measured: persistent is slower there)
bool lde_persistent_supported(u32 BT) { return BT == 11; }

template <int BT>
static int launch_top_p(Ctx* ctx, const LdeParams& p, bool dit)
{
    constexpr u32 V = 8;
    const u32 ncc = (p.wc + V - 1) / V;
    const size_t lds = ((size_t)((1u << BT) + (1u << BT) / 16 + 1) * V + (1u << BT)) * sizeof(u64);
    static bool attr_done = false;
    if (!attr_done) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_top_dif_p<BT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        XLP_HIP(hipFuncSetAttribute((const void*)lde_top_dit_p<BT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const u64 ntiles = dit ? ((u64)ncc << 12) << p.e : (u64)ncc << 12;
    const u32 grid = ntiles < ctx->ncu ? (u32)((ntiles + 7) / 8 * 8) : ctx->ncu;
    if (!dit) hipLaunchKernelGGL((lde_top_dif_p<BT, true>), dim3(grid), dim3(PT), lds, ctx->stream, p, (u32)ntiles);
    else hipLaunchKernelGGL(lde_top_dit_p<BT>, dim3(grid), dim3(PT), lds, ctx->stream, p, (u32)ntiles);
    XLP_HIP(hipGetLastError());
    return 0;
}

int launch_lde_top_p(Ctx* ctx, const LdeParams& p, u32 BT, bool dit)
{
    switch (BT) {
        case 10: return launch_top_p<10>(ctx, p, dit);
        case 11: return launch_top_p<11>(ctx, p, dit);
        default: return xlp_fail("persistent lde: unsupported size");
    }
}

int launch_lde_mid_p(Ctx* ctx, const LdeParams& p, u32 BT)
{
    const u32 ncc = (p.wc + 3) / 4;
    const u64 ntiles = (u64)ncc << BT;
    const size_t lds = (size_t)(4096 + 256 + 1) * 4 * sizeof(u64);
    static bool attr_done = false;
    if (!attr_done) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_mid_p, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const u32 grid = ntiles < ctx->ncu ? (u32)((ntiles + 7) / 8 * 8) : ctx->ncu;
    hipLaunchKernelGGL(lde_mid_p, dim3(grid), dim3(PT), lds, ctx->stream, p, (u32)ntiles);
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
