// K2 of the extendPol fast path (see ntt_lde.hip): the 12-bit middle pass.  A translation unit of its own because it is compiled with
// -mllvm -amdgpu-sched-strategy=max-ilp (csrc/Makefile): the kernel is integer-VALU bound with 128 VGPRs and four waves per SIMD, and the default
// (occupancy-first) list scheduler leaves 1 366 hazard wait states between carry producers and consumers in its body; the ILP-first strategy fills 40 %
// of them with independent butterflies (857) and spills less (14 instead of 23 registers): 1.19 -> 1.12 ms at 2^23 x 8
// (profiles/r3/lde_sched_strategy_ab.txt).  The same flag does nothing for lde_top_dif and costs lde_top_dit 1-2 %, so only this kernel gets it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "ntt_common.hpp"
#include "ntt_lde_common.hpp"

namespace xlp {

// ------------------------------------------------------------------------ K2
__global__ __launch_bounds__(XLP_WGT) void lde_mid(const LdeParams p, u32 BT)
{
    constexpr int V = 4, B = 12;
    extern __shared__ u64 sm[];
    const u32 ncc = (p.wc + V - 1) / V;
    const u32 tile = xcd_tile();
    const u32 cc = tile % ncc, hi = tile / ncc;
    const u32 c = cc * V;
    const u64 n = 1ull << p.L;
    const u64 goff = (((u64)(cc >> 1) << p.L) << 3) + (cc & 1) * 4;        // my 4 columns inside their 8-column group of a scratch matrix (lde_sidx)
    const u64* __restrict__ Ag = p.A + goff;
    auto gload = [&](u32 tpos, u32 v, u32) -> u64 {
        const u32 vv = c + v < p.wc ? v : p.wc - 1 - c;
        const u64 row = ((u64)hi << 12) | tpos;
        return gld<2>(&Ag[row * 8 + vv]);
    };
    auto lload = [&](u32, u32 v, u32 lrow) -> u64 { return sm[lrow * V + v]; };
    auto lstore = [&](u32, u32 v, u64 val, u32 lrow) { sm[lrow * V + v] = val; };
    // ---- inverse NTT over the low 12 bits (DIF); last round keeps its outputs in VGPRs
    fast_round<B, V, 8, 4, false, true>(p.twi_12, gload, lstore);
    XLP_SYNC();
    fast_round<B, V, 4, 4, false, true>(p.twi_12, lload, lstore);
    XLP_SYNC();
    // a thread's groups in a TP = 0 round: g = tid + i * T -> base = (g >> 2) << 4, v = g & 3  (T % 4 == 0)
    constexpr int NGT = 1024 / XLP_WGT;
    const u32 v = threadIdx.x & 3;
    u32 base_[NGT];
    u64 X[NGT][16];
#pragma unroll
    for (int i = 0; i < NGT; i++) {
        base_[i] = ((threadIdx.x + i * XLP_WGT) >> 2) << 4;
        const u32 base = base_[i];
        u64 x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = sm[(lds_row(base) + j) * V + v];
        dft_regs<4>(x);
#pragma unroll
        for (int k = 0; k < 16; k++) X[i][k] = x[brev<4>((16 - k) % 16)];   // coefficient at tile position base + k
    }
    const bool valid = c + v < p.wc;
    // ---- every coset: scale, forward NTT over the same 12 bits (DIT), -> B_r
    // Coset scale S_r[k2] = G^k2 (G = g_r^(2^BT), k2 = natural index of the tile position):
    // for the 16 inputs of one thread k2 = nb + 256 j, so S = G^nb * H[j], H[j] = G^(256 j)
    // workgroup-uniform.  H[j] is applied here; G^nb is constant over a thread's DFT16, commutes
    // with it, and is folded into the position-indexed twiddle table tw2[r] of the next round.
    for (u32 r = 0; r < (1u << p.e); r++) {
        const u64* __restrict__ hk = p.hk[r];
        u64* __restrict__ Br = p.B + (u64)r * lde_swords(p.wc, p.L) + goff;
        XLP_SYNC();   // previous users of the LDS tile are done
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
        XLP_SYNC();
        fast_round<B, V, 4, 4, true, false, false, false, decltype(lload), decltype(lstore), NoFactor, NoFactor, true>(p.tw2[r], lload, lstore, NoFactor{}, NoFactor{});
        XLP_SYNC();
        auto gstore = [&](u32 tpos, u32 vv, u64 val, u32) {
            const u64 row = ((u64)hi << 12) | tpos;
            if (c + vv < p.wc) gst<3>(&Br[row * 8 + vv], val);
        };
        fast_round<B, V, 8, 4, true, false>(p.twf_12, lload, gstore);
    }
    (void)valid; (void)BT;
}

int launch_lde_mid(Ctx* ctx, const LdeParams& p, u32 BT)
{
    static bool mid_attr = false;
    if (!mid_attr) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_mid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        mid_attr = true;
    }
    const u32 ncc = (p.wc + 3) / 4;
    const u64 grid = (u64)ncc << BT;
    const size_t lds = (size_t)(4096 + 256 + 1) * 4 * sizeof(u64);
    hipLaunchKernelGGL(lde_mid, dim3((unsigned)grid), dim3(XLP_WGT), lds, ctx->stream, p, BT);
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
