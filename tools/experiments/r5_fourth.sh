#!/bin/bash
# Round 5, fourth GPU call: the whole GPU suite on the rebuilt product (plugins generated with --hoist-loads 4 --cut 320, canonical sections, lazy canonicalisation),
# with the slowest tests named, then the bench line.
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
python3 -m pytest tests -q -m gpu -x --durations=60 > $R/gputest_full_1.txt 2>&1
echo "pytest rc $?"
tail -5 $R/gputest_full_1.txt
python3 bench.py > $R/bench_mid.json 2> $R/bench_mid.err
echo "bench rc $?"
