// CPU stand-in for the few HIP constructs the generated expression kernels use (tests only: lets the
// generated straight-line code run under the host compiler + sanitizers against the oracle).
#pragma once
#include <cstdint>
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static dim3 blockIdx, threadIdx;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
static inline int hipGetLastError() { return hipSuccess; }
#define __global__
#define __device__
#define __forceinline__ inline
#define __launch_bounds__(x)
#define __restrict__
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)      /* a scheduling fence of the GPU compiler (xlayer_prover_amd/expr_hoist.py) */
#define hipLaunchKernelGGL(k, g, b, sh, st, ...)                                   \
    for (unsigned bx_ = 0; bx_ < (g).x; bx_++)                                     \
        for (unsigned tx_ = 0; tx_ < (b).x; tx_++) { blockIdx.x = bx_; threadIdx.x = tx_; k(__VA_ARGS__); }
