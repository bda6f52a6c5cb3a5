// Shared declarations of the column-major extendPol kernels (ntt_lde_cm.hip: the two top passes; ntt_lde_cm_mid.hip: the middle pass, a
// translation unit of its own because it takes another instruction-scheduling strategy, like its row-major twin: csrc/Makefile).
#pragma once
#include "ntt_lde_common.hpp"

namespace xlp {

struct LdeCmParams {
    const u64* in; u64 in_cs;            // input: column c at in + c * in_cs, n words
    u64* out; u64 out_cs, out_cos;       // output: column c at out + c * out_cs, coset r at + r * out_cos, n words each
    u64* A;                              // scratch [wc][n]
    u64* B;                              // scratch [2^e][wc][n]
    u32 col0, wc;                        // column chunk [col0, col0 + wc)
    u32 L, e;
    const u64 *twi_bt, *twf_bt, *twi_12, *twf_12;
    const u64 *cri_lo, *cri_hi, *crf_lo, *crf_hi;
    const u64 *ei, *ef;                  // ei[k][v] = w(L)^(-k v), ef[k][v] = w(L)^(k v): v < 8, k < 2^BT
    const u64* tw2[LDE_MAX_COSETS];
    const u64* hk[LDE_MAX_COSETS];
    const u64* ur[LDE_MAX_COSETS];
};

int launch_lde_cm_mid(Ctx* ctx, const LdeCmParams& p, u32 BT);

}  // namespace xlp
