#!/bin/bash
# Round 5, fifth GPU call: the evidence run at HEAD (kernel stats, counters, per-class counters: tools/experiments/r5_profile_run.sh), then the thinned GPU suite timed
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
R=gpurun_out/r5
mkdir -p $R
bash tools/experiments/r5_profile_run.sh > $R/profile_run.log 2>&1
echo "profile rc $?"; tail -3 $R/profile_run.log
python3 -m pytest tests -q -m gpu -x --durations=25 > $R/gputest_full_2.txt 2>&1
echo "pytest rc $?"
tail -4 $R/gputest_full_2.txt
