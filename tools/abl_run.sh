#!/bin/bash
# runs bench under rocprofv3 kernel-trace for each ablation lib; prints per-kernel average ns
cd /tmp; export TMPDIR=/tmp
for n in 0 "$@"; do
  if [ "$n" = "0" ]; then unset XLP_LIB; else export XLP_LIB=$GRAFT_REPO_ROOT/build/abl$n/libxlp_gl.so; fi
  rm -rf /tmp/abl_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-genproof > /tmp/abl.log 2>&1
  echo "== ABL $n"
  grep -h "lde_" /tmp/abl_prof/*/*kernel_stats.csv | awk -F, '{printf "%-50s %10.1f us\n", $1, $4/1000}'
done
