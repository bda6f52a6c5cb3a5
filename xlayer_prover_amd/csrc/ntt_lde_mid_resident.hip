// The RESIDENT form of K2 of the extendPol fast path (ntt_lde_common.hpp; xlp_dev_lde_resident, the ALU ceiling of bench.py's headline): the kernel text
// of ntt_lde_mid_kernel.inc with its global data traffic removed.  A translation unit of its own, compiled with the same flags as ntt_lde_mid.hip
// (csrc/Makefile): a second kernel in the product's unit changes what the inliner and the max-ilp scheduler make of the first.
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

#define LDE_MID_KERNEL lde_mid_resident
#define LDE_MID_RES 1
#include "ntt_lde_mid_kernel.inc"
#undef LDE_MID_KERNEL
#undef LDE_MID_RES

int launch_lde_mid_resident(Ctx* ctx, const LdeParams& p, u32 BT)
{
    static bool attr = false;
    if (!attr) {
        XLP_HIP(hipFuncSetAttribute((const void*)lde_mid_resident, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const u32 ncc = (p.wc + LDE_MID_V - 1) / LDE_MID_V;
    const u64 grid = (u64)ncc << BT;
    const size_t lds = (size_t)(4096 + 256 + 1) * LDE_MID_V * sizeof(u64);
    hipLaunchKernelGGL(lde_mid_resident, dim3((unsigned)grid), dim3(XLP_WGT), lds, ctx->stream, p, BT);
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
