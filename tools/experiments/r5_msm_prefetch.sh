#!/bin/bash
# NEXT ROUND (prepared at the end of round 4, not run on a GPU): the bucket pass of xlp_dev_msm_bn254 with the NEXT point of a bucket loaded while
# the current one is added.  tools/experiments/isa_loads_in_flight.py: msm_buckets_kernel waits for each gathered point (index, then 64 bytes) with
# nothing else in flight.  The same idea cost 2 % in the middle of round 4 -- under the old call structure (operands through scratch, 196 registers);
# the kernel now has no calls and 128 registers.  Builds gpurun_ab/libxlp_gl.so; A/B on ONE box:
#   xlayer_prover_amd/host/test_host_shim fflonk_bench 22 3      against      LD_LIBRARY_PATH=$PWD/gpurun_ab xlayer_prover_amd/host/test_host_shim fflonk_bench 22 3
# and  XLP_LIB=$PWD/gpurun_ab/libxlp_gl.so python -m pytest tests/test_bn128.py -q -m gpu -k msm
set -e
cd "$(dirname "$0")/../../xlayer_prover_amd/csrc"
python3 - <<'PY'
s = open("bn254_msm.hip").read()
old = "        for (u32 k = b; k < e; k++) acc = jmadd_inl(acc, bases[sorted_vals[k]]);"
new = '''        Aff nxt = bases[sorted_vals[b]];
        for (u32 k = b; k < e; k++) {
            const Aff cur = nxt;
            if (k + 1 < e) nxt = bases[sorted_vals[k + 1]];          // in flight during the addition below
            acc = jmadd_inl(acc, cur);
        }'''
assert s.count(old) == 1
open("bn254_msm_prefetch_tmp.hip", "w").write(s.replace(old, new))
PY
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -c bn254_msm_prefetch_tmp.hip -o /tmp/bn254_msm_prefetch.o
rm bn254_msm_prefetch_tmp.hip
mkdir -p ../../gpurun_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_ab/libxlp_gl.so $(ls *.o | grep -v '^bn254_msm.o$') /tmp/bn254_msm_prefetch.o -ldl -lrt -lpthread
echo built gpurun_ab/libxlp_gl.so
