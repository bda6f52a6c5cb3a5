#!/bin/bash
# round 4: column-major sections through the C++ prover -- parity tests, then the batch proof at full size in both layouts, then the
# VALU / MFMA instruction counters of the two MDS forms (tools/ubench_mds_mfma.hip)
mkdir -p gpurun_out/r4
timeout -k 10 800 python -m pytest tests/test_colmajor_kernels.py tests/test_cpp_colmajor.py -m gpu -x -q -k "evmap or dot_columns or cpp or colmajor_zkevm or small_starks or genbatchproof" > gpurun_out/r4/t_cm2.log 2>&1
echo "cm2 rc $?" >> gpurun_out/r4/t_cm2.log
tail -30 gpurun_out/r4/t_cm2.log
for c in synthetic synthetic_colmajor; do
  timeout -k 10 200 xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_$c.json --proofs 2 --timers > gpurun_out/r4/batch_$c.json 2> gpurun_out/r4/batch_$c.err
  echo "$c rc $?"
  tail -c 300 gpurun_out/r4/batch_$c.json
done
cd /tmp && export TMPDIR=/tmp
for pmc in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/r4/mds_pmc_$pmc -o mds -- $GRAFT_REPO_ROOT/build/ubench_mds_mfma 20 64 > $GRAFT_REPO_ROOT/gpurun_out/r4/mds_pmc_$pmc.log 2>&1
  echo "pmc $pmc rc $?"
done
