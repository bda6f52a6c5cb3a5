#!/bin/bash
# Round 5, second half: evidence at HEAD with the matrix-pipe middle pass as the product kernel -- kernel stats + counters (r5_profile_run.sh), then the bench line
# (the suite was run by itself: profiles/r5/gputest_r5_mfma.log)
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
R=gpurun_out/r5
mkdir -p $R
bash tools/experiments/r5_profile_run.sh > $R/profile_run.log 2>&1
echo "profile rc $?"; tail -2 $R/profile_run.log
python3 bench.py > $R/bench_mfma.json 2> $R/bench_mfma.err
echo "bench rc $?"
tail -c 1500 $R/bench_mfma.json
