#!/usr/bin/env python3
"""Generates + builds build/ubench3: per-instruction throughput on gfx950 (volatile inline asm, 4 waves/SIMD)."""
import subprocess, os
tests = {
 "v_add_u32":        "v_add_u32 {d}, {a}, {b}",
 "v_add_co_u32 vcc": "v_add_co_u32 {d}, vcc, {a}, {b}",
 "v_add_co_u32 sgpr":"v_add_co_u32 {d}, s[20:21], {a}, {b}",
 "v_addc_co_u32 vcc":"v_addc_co_u32 {d}, vcc, {a}, {b}, vcc",
 "v_addc_co sgpr":   "v_addc_co_u32 {d}, s[20:21], {a}, {b}, s[22:23]",
 "v_cndmask vcc":    "v_cndmask_b32 {d}, {a}, {b}, vcc",
 "v_cndmask sgpr":   "v_cndmask_b32 {d}, {a}, {b}, s[22:23]",
 "v_lshl_add_u64":   "v_lshl_add_u64 {D}, {A}, 0, {B}",
 "v_cmp_lt_u64":     "v_cmp_lt_u64 s[20:21], {A}, {B}",
 "v_cmp_lt_u32":     "v_cmp_lt_u32 s[20:21], {a}, {b}",
 "v_mad_u64_u32":    "v_mad_u64_u32 {D}, s[20:21], {a}, {b}, {A}",
 "v_mul_lo_u32":     "v_mul_lo_u32 {d}, {a}, {b}",
 "v_mul_hi_u32":     "v_mul_hi_u32 {d}, {a}, {b}",
 "v_alignbit_b32":   "v_alignbit_b32 {d}, {a}, {b}, 12",
 "v_add3_u32":       "v_add3_u32 {d}, {a}, {b}, {a}",
 "v_lshlrev_b64":    "v_lshlrev_b64 {D}, 12, {A}",
 "v_min_u32":        "v_min_u32 {d}, {a}, {b}",
 "v_sub_co_u32 vcc": "v_sub_co_u32 {d}, vcc, {a}, {b}",
 "v_xor_b32":        "v_xor_b32 {d}, {a}, {b}",
 "v_mad_u32_u24":    "v_mad_u32_u24 {d}, {a}, {b}, {a}",
 "v_perm_b32":       "v_perm_b32 {d}, {a}, {b}, {a}",
 "v_bfe_u32":        "v_bfe_u32 {d}, {a}, 4, 12",
 "v_mov_b32":        "v_mov_b32 {d}, {a}",
 "v_pk_add_u16":     "v_pk_add_u16 {d}, {a}, {b}",
 "v_mad_i32_i24":    "v_mad_i32_i24 {d}, {a}, {b}, {a}",
 "v_mul_u32_u24":    "v_mul_u32_u24 {d}, {a}, {b}",
 "v_dot4_u32_u8":    "v_dot4_u32_u8 {d}, {a}, {b}, {a}",
 "v_mad_u64_u32 dep":"v_mad_u64_u32 v[40:41], s[20:21], {a}, {b}, v[40:41]",
}
N = 32
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>',
       '#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\\n", hipGetErrorString(e)); exit(1);} } while (0)',
       'constexpr int ITER = 4096;']
names = list(tests)
for ti, name in enumerate(names):
    tpl = tests[name]
    lines = []
    for i in range(N):
        r = 2 * (i % 8)
        lines.append(tpl.format(d=f"v{r+10}", a=f"v{(r+2)%16+10}", b=f"v{(r+4)%16+10}", D=f"v[{r+10}:{r+11}]", A=f"v[{(r+2)%16+10}:{(r+2)%16+11}]", B=f"v[{(r+4)%16+10}:{(r+4)%16+11}]"))
    body = "\\n\\t".join(lines)
    clob = ", ".join(f'"v{i}"' for i in range(10, 42)) + ', "vcc", "s20", "s21", "s22", "s23"'
    src.append(f'__global__ __launch_bounds__(256) void k{ti}(unsigned* o) {{ for (int it = 0; it < ITER; it++) asm volatile("{body}" ::: {clob}); if (o) o[threadIdx.x] = 0; }}')
src.append('template <class K> void run(K kern, const char* name) { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); int blocks = 256 * 4; hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (unsigned*)nullptr); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int r = 0; r < 3; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (unsigned*)nullptr); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3; double ops = (double)blocks * 256 * ITER * %d; printf("%%-22s %%8.3f ms  %%6.2f lane-cycles/instr @2.4GHz (%%.2f SIMD-cycles per wave-instr)\\n", name, ms, (256.0 * 4 * 32 * 2.4e9) / (ops / (ms * 1e-3)), (256.0 * 4 * 32 * 2.4e9) / (ops / (ms * 1e-3)) * 2 / 1.0 / 1.0); }' % N)
src.append('int main() {')
for ti, name in enumerate(names):
    src.append(f'  run(k{ti}, "{name}");')
src.append('  return 0; }')
os.makedirs("build", exist_ok=True)
open("build/ubench3.hip", "w").write("\n".join(src))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", "build/ubench3", "build/ubench3.hip"])
print("built build/ubench3")
