#!/bin/bash
mkdir -p gpurun_out/r4
timeout -k 10 700 python -m pytest tests/test_colmajor_kernels.py tests/test_cpp_colmajor.py tests/test_expr_colmajor.py -m gpu -q > gpurun_out/r4/t_cm3.log 2>&1
echo "cm3 rc $?" >> gpurun_out/r4/t_cm3.log
tail -25 gpurun_out/r4/t_cm3.log
timeout -k 10 200 python tools/experiments/bench_lde_cm.py > gpurun_out/r4/lde_cm_ab.json 2> gpurun_out/r4/lde_cm_ab.err; echo "lde ab rc $?"; cat gpurun_out/r4/lde_cm_ab.json
for c in synthetic synthetic_colmajor; do
  timeout -k 10 200 xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_$c.json --proofs 2 --timers > gpurun_out/r4/batch2_$c.json 2> gpurun_out/r4/batch2_$c.err
  echo "$c rc $?"
  tail -c 250 gpurun_out/r4/batch2_$c.json
done
