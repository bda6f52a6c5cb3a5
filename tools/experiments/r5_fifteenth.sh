#!/bin/bash
# Round 5, fifteenth GPU call: counters / kernel stats re-collected on the final tree (the hash of the measured sources now covers ntt_lde_mid_kernel.inc), then the bench line
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
R=gpurun_out/r5
mkdir -p $R
bash tools/experiments/r5_profile_run.sh > $R/profile_run.log 2>&1
echo "profile rc $?"; tail -2 $R/profile_run.log
cp $R/counters.json profiles/r5/counters.json     # (so that the bench line below reads the counters of THIS tree)
python3 bench.py > $R/bench_last.json 2> $R/bench_last.err
echo "bench rc $?"
