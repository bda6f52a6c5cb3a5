#!/bin/bash
# Builds ablation variants of libxlp_gl.so (XLP_ABL=n in ntt_lde.hip) into build/abl<n>/ .
set -e
cd "$(dirname "$0")/../xlayer_prover_amd/csrc"
for n in "$@"; do
  d=../../build/abl$n; mkdir -p $d
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $( [ $n -ge 1000 ] && echo -DXLP_NT=$((n-1000)) || ( [ $n -lt 100 ] && echo -DXLP_ABL=$n || ( [ $n -eq 100 ] && echo -DXLP_TOPV=4 || echo -DXLP_WGT=$n ) ) ) -c ntt_lde.hip -o $d/ntt_lde.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libxlp_gl.so ntt.o $d/ntt_lde.o poseidon.o fri.o polinomial.o expr.o xlp_capi.o
done
