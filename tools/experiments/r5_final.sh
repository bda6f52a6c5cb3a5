#!/bin/bash
# Round 5, the evidence run at HEAD: counters / kernel stats (r5_profile_run.sh), the GPU suite, the bench line
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
R=gpurun_out/r5
mkdir -p $R
bash tools/experiments/r5_profile_run.sh > $R/profile_run.log 2>&1
echo "profile rc $?"; tail -2 $R/profile_run.log
python3 -m pytest tests -q -m gpu -x --durations=15 > $R/gputest_final.txt 2>&1
echo "pytest rc $?"
tail -3 $R/gputest_final.txt
python3 bench.py > $R/bench_final.json 2> $R/bench_final.err
echo "bench rc $?"
