// K2 of the column-major extendPol (see ntt_lde_cm.hip): the 12-bit middle pass.  Compiled with -mllvm -amdgpu-sched-strategy=max-ilp like
// lde_mid (csrc/Makefile): integer-VALU bound, four waves per SIMD, carry chains that want independent butterflies between producer and consumer.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "ntt_common.hpp"
#include "ntt_lde_common.hpp"
#include "ntt_lde_cm_common.hpp"

namespace xlp {

// ------------------------------------------------------------------------ K2: the 12-bit middle pass on 4 whole columns
__global__ __launch_bounds__(1024) void lde_cm_mid(const LdeCmParams p, u32 BT)
{
    constexpr int V = 4, B = 12;
    constexpr u32 LT = 4096 + 256 + 1;                       // padded tile of one column
    extern __shared__ u64 sm[];
    const u32 ncc = (p.wc + V - 1) / V;
    const u32 tile = xcd_tile();
    const u32 cc = tile % ncc, hi = tile / ncc;
    const u32 c = cc * V;
    const u64 n = 1ull << p.L;
    // columns past the chunk's last read its last column (in bounds) and never store
    auto colbase = [&](u32 v) -> u64 { return (u64)(c + v < p.wc ? c + v : p.wc - 1) << p.L; };
    auto gload = [&](u32 tpos, u32 v, u32) -> u64 { return p.A[colbase(v) + (((u64)hi << 12) | tpos)]; };
    auto lload = [&](u32, u32 v, u32 lrow) -> u64 { return sm[v * LT + lrow]; };
    auto lstore = [&](u32, u32 v, u64 val, u32 lrow) { sm[v * LT + lrow] = val; };
    fast_round<B, V, 8, 4, false, true, false, false, decltype(gload), decltype(lstore), NoFactor, NoFactor, false, true>(p.twi_12, gload, lstore, NoFactor{}, NoFactor{});
    XLP_SYNC();
    fast_round<B, V, 4, 4, false, true, false, false, decltype(lload), decltype(lstore), NoFactor, NoFactor, false, true>(p.twi_12, lload, lstore, NoFactor{}, NoFactor{});
    XLP_SYNC();
    // the last inverse round keeps its outputs in registers: thread -> (column v = tid >> 8, group = tid & 255)
    const u32 v = threadIdx.x >> 8;
    const u32 base = (threadIdx.x & 255) << 4;
    u64 X[16];
    {
        u64 x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = sm[v * LT + lds_row(base) + j];
        dft_regs<4>(x);
#pragma unroll
        for (int k = 0; k < 16; k++) X[k] = x[brev<4>((16 - k) % 16)];
    }
    const bool valid = c + v < p.wc;
    for (u32 r = 0; r < (1u << p.e); r++) {
        const u64* __restrict__ hk = p.hk[r];
        u64* __restrict__ Br = p.B + (((u64)r * p.wc) << p.L);
        XLP_SYNC();
        {
            u64 x[16];
            x[0] = X[0];
#pragma unroll
            for (int j = 1; j < 16; j++) x[j] = gl::mulc(X[j], hk[j]);
            dft_regs<4>(x);
#pragma unroll
            for (int k = 0; k < 16; k++) sm[v * LT + lds_row(base) + k] = x[brev<4>(k)];
        }
        XLP_SYNC();
        fast_round<B, V, 4, 4, true, false, false, false, decltype(lload), decltype(lstore), NoFactor, NoFactor, true, true>(p.tw2[r], lload, lstore, NoFactor{}, NoFactor{});
        XLP_SYNC();
        auto gstore = [&](u32 tpos, u32 vv, u64 val, u32) {
            if (c + vv < p.wc) Br[((u64)(c + vv) << p.L) + (((u64)hi << 12) | tpos)] = val;
        };
        fast_round<B, V, 8, 4, true, false, false, false, decltype(lload), decltype(gstore), NoFactor, NoFactor, false, true>(p.twf_12, lload, gstore, NoFactor{}, NoFactor{});
    }
    (void)valid; (void)BT; (void)n;
}

int launch_lde_cm_mid(Ctx* ctx, const LdeCmParams& p, u32 BT)
{
    static bool mid_attr = false;
    if (!mid_attr) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_cm_mid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        mid_attr = true;
    }
    const u32 ncc = (p.wc + 3) / 4;
    const size_t lds = (size_t)(4096 + 256 + 1) * 4 * sizeof(u64);
    hipLaunchKernelGGL(lde_cm_mid, dim3((unsigned)((u64)ncc << BT)), dim3(1024), lds, ctx->stream, p, BT);
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
