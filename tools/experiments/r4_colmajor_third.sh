#!/bin/bash
mkdir -p gpurun_out/r4
timeout -k 10 500 python -m pytest tests/test_colmajor_kernels.py tests/test_cpp_colmajor.py -m gpu -q -k "not small_starks" > gpurun_out/r4/t_cm4.log 2>&1
echo "cm4 rc $?" >> gpurun_out/r4/t_cm4.log
tail -8 gpurun_out/r4/t_cm4.log
timeout -k 10 200 python tools/experiments/bench_lde_cm.py > gpurun_out/r4/lde_cm_ab2.json 2> gpurun_out/r4/lde_cm_ab2.err; echo "lde ab rc $?"; cat gpurun_out/r4/lde_cm_ab2.json
timeout -k 10 420 python bench.py --genproof-rowmajor > gpurun_out/r4/bench_mid.json 2> gpurun_out/r4/bench_mid.err; echo "bench rc $?"; tail -c 1500 gpurun_out/r4/bench_mid.json
