#!/bin/bash
# NEXT ROUND (prepared at the end of round 4, not run on a GPU): the fused partial-round constants of the Poseidon kernels by SCALAR loads.
# csrc/poseidon.hip partial3 launders its table pointer through an SGPR asm constraint; that also loses the constant address space, and the 193
# constants arrive by 49 flat_load_dwordx4 of a uniform address per call (7 calls per permutation) into vector registers.  This builds a second
# library with the pointer typed __attribute__((address_space(4))) (CPU check at the end of round 4: flat 49 -> 0, s_load 45 -> 77, registers
# 76 -> 59, vector instructions +1 % static) into gpurun_ab/libxlp_gl.so.  A/B on ONE box:
#   python bench.py --no-e2e --no-genproof --no-bn128              (roofline_merkle: sponge and alu_ceiling)
#   XLP_LIB=$PWD/gpurun_ab/libxlp_gl.so python bench.py --no-e2e --no-genproof --no-bn128
# and the Poseidon known answers on the variant:  XLP_LIB=... python -m pytest tests/test_gpu_parity.py -q -m gpu -k "poseidon or merkle"
set -e
cd "$(dirname "$0")/../../xlayer_prover_amd/csrc"
python3 - <<'PY'
s = open("poseidon.hip").read()
old = '''    const u32* F = POSEIDON_FU32;
    asm volatile("" : "+s"(F));
    const u32* __restrict__ R1 = F;             // row 0 of M
    const u32* __restrict__ R2 = F + 12;        // row 0 of M Z M
    const u32 m00 = F[24];
    const u32* __restrict__ B3 = F + 25;        // M Z M Z M, row-major
    const u32* __restrict__ B2 = F + 25 + 144;  // M Z M e0
    const u32* __restrict__ A1 = F + 25 + 156;  // M e0'''
new = '''    typedef const __attribute__((address_space(4))) u32* cu32p;
    cu32p F = (cu32p)POSEIDON_FU32;
    asm volatile("" : "+s"(F));
    cu32p R1 = F;             // row 0 of M
    cu32p R2 = F + 12;        // row 0 of M Z M
    const u32 m00 = F[24];
    cu32p B3 = F + 25;        // M Z M Z M, row-major
    cu32p B2 = F + 25 + 144;  // M Z M e0
    cu32p A1 = F + 25 + 156;  // M e0'''
assert old in s
open("poseidon_as4_tmp.hip", "w").write(s.replace(old, new))
PY
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-sched-strategy=iterative-minreg -c poseidon_as4_tmp.hip -o /tmp/poseidon_as4.o
rm poseidon_as4_tmp.hip
mkdir -p ../../gpurun_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_ab/libxlp_gl.so $(ls *.o | grep -v '^poseidon.o$') /tmp/poseidon_as4.o -ldl -lrt -lpthread
echo built gpurun_ab/libxlp_gl.so
