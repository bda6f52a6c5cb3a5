#!/bin/bash
# Round 5, A/B: lde_top_dit with the first round's inputs requested before the cross vector is built (-DXLP_TOP_PRELOAD=1 variant library in
# gpurun_ab/top_preload) against the product, alternated; parity of the variant first (barycentric + oracle LDE tests through XLP_LIB).
set -e
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
V=$PWD/gpurun_ab/top_preload/libxlp_gl.so
[ -n "$SKIP_PARITY" ] || XLP_LIB=$V timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_colmajor_kernels.py -m gpu -x -q -k "lde or extend or bary" > $R/top_preload_parity.txt 2>&1
[ -n "$SKIP_PARITY" ] || tail -2 $R/top_preload_parity.txt
: > $R/top_preload_ab.txt
for i in 1 2 3; do
  for lib in product variant; do
    if [ $lib = variant ]; then export XLP_LIB=$V; else unset XLP_LIB; fi
    timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-e2e --no-genproof --no-cpu-baseline --no-bn128 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); r=j['roofline']
print('$lib', j['value'], r.get('avg_launch_ms'))" >> $R/top_preload_ab.txt
    tail -1 $R/top_preload_ab.txt
  done
done
