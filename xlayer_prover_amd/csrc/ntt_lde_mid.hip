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

#ifndef LDE_MID_V
#define LDE_MID_V 4
#endif
namespace xlp {

// ------------------------------------------------------------------------ K2
#define LDE_MID_KERNEL lde_mid
#define LDE_MID_RES 0
#include "ntt_lde_mid_kernel.inc"
#undef LDE_MID_KERNEL
#undef LDE_MID_RES

int launch_lde_mid_resident(Ctx* ctx, const LdeParams& p, u32 BT);   // ntt_lde_mid_resident.hip
int launch_lde_mid(Ctx* ctx, const LdeParams& p, u32 BT, bool resident)
{
    if (resident) return launch_lde_mid_resident(ctx, p, BT);
    static bool mid_attr = false;
    if (!mid_attr) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_mid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        mid_attr = true;
    }
    const u32 ncc = (p.wc + LDE_MID_V - 1) / LDE_MID_V;
    const u64 grid = (u64)ncc << BT;
    const size_t lds = (size_t)(4096 + 256 + 1) * LDE_MID_V * sizeof(u64);
    hipLaunchKernelGGL(lde_mid, dim3((unsigned)grid), dim3(XLP_WGT), lds, ctx->stream, p, BT);
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
