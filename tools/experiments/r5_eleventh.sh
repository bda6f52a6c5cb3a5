#!/bin/bash
# Round 5, eleventh GPU call: the GPU suite timed once more (the 2^8 plugins prebuilt), and the --gpus 2 path of bench.py rehearsed with the two ranks SHARING this box's
# GPU (gloo / shm: plumbing evidence, the rates mean nothing)
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
python3 -m pytest tests -q -m gpu -x --durations=12 > $R/gputest_last.txt 2>&1
echo "pytest rc $?"
tail -3 $R/gputest_last.txt
timeout -k 10 420 python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 > $R/bench_gpus2_rehearsal.json 2> $R/bench_gpus2_rehearsal.err
echo "rehearsal rc $?"
tail -c 1500 $R/bench_gpus2_rehearsal.json
