#!/bin/bash
# bench.py (its own HIP-event per-kernel timing) for each ablation library build/abl<n>; 0 = the product
for n in 0 "$@"; do
  if [ "$n" = "0" ]; then unset XLP_LIB; else export XLP_LIB=$GRAFT_REPO_ROOT/build/abl$n/libxlp_gl.so; fi
  python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-genproof 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('ABL $n', d['ms_per_step'], {k.split('::')[-1]: v for k, v in d['roofline']['avg_launch_ms'].items()})"
done
