#!/bin/bash
# Builds ablation variants of libxlp_gl.so into build/abl<n>/ from tools/experiments/ntt_lde_ablation.hip: the LDE kernels of round 1 WITH their
# measurement switches (XLP_ABL: no twiddle loads / no global traffic / no DFT / ..., XLP_NT: non-temporal accesses).  The product kernels
# (xlayer_prover_amd/csrc/ntt_lde.hip) carry none of these; profiles/r1/ablation_lde_kernels.txt was measured with this file.
set -e
cd "$(dirname "$0")/../xlayer_prover_amd/csrc"
for n in "$@"; do
  d=../../build/abl$n; mkdir -p $d
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $( [ $n -ge 1000 ] && echo -DXLP_NT=$((n-1000)) || ( [ $n -lt 100 ] && echo -DXLP_ABL=$n || ( [ $n -eq 100 ] && echo -DXLP_TOPV=4 || echo -DXLP_WGT=$n ) ) ) -I . -I ../../tools/experiments -c ../../tools/experiments/ntt_lde_ablation.hip -o $d/ntt_lde.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libxlp_gl.so ntt.o $d/ntt_lde.o poseidon.o fri.o polinomial.o expr.o xlp_capi.o
done
