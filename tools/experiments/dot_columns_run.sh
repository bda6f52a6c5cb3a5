# dot_columns after a change (GPU box): the suites that run it, then kernel-trace stats of one batch proof (the zkEVM launch is the long one)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_expr_vm.py tests/test_cpp_starks_zkevm.py tests/test_cpp_sharded.py -x -q -m gpu > $O/dot_columns_suites.log 2>&1 || { tail -30 $O/dot_columns_suites.log; exit 1; }
tail -2 $O/dot_columns_suites.log
rm -rf $O/prof_dc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dc -- $R/xlayer_prover_amd/host/zkprover_batch -c $R/tools/config_runFile_BatchProof_synthetic.json --proofs 1 > $O/prof_dc.out 2> $O/prof_dc.err
cp $(ls $O/prof_dc/*/*kernel_stats.csv | head -1) $O/dot_columns_kernel_stats.csv
grep -E "dot_columns|evmap_partial" $O/dot_columns_kernel_stats.csv
python3 -c "
import json,sys
j=json.loads(open('$O/prof_dc.out').read().strip().splitlines()[-1])
print(j['batch_proof_fnv'], j['timers_s']['zkevm.STARK_STEP_5_CALCULATE_EXPS'], j['three_starks_s'])"
