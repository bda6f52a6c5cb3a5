#!/bin/bash
# Round 5, twelfth GPU call: what the driver runs at round end, on the final tree: smoke(), the GPU suite, the bench line
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
python3 -c "import __graft_entry__ as g; g.smoke()" > $R/smoke_final.txt 2>&1
echo "smoke rc $?"; tail -2 $R/smoke_final.txt
python3 -m pytest tests -q -m gpu -x --durations=8 > $R/gputest_confirm.txt 2>&1
echo "pytest rc $?"
tail -3 $R/gputest_confirm.txt
python3 bench.py > $R/bench_confirm.json 2> $R/bench_confirm.err
echo "bench rc $?"
