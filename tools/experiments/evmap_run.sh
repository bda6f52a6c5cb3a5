# evmap after a change (GPU box): parity tests, then the kernel inside one batch proof (kernel-trace stats)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "evmap or dot_columns" 2>&1 | tail -2
rm -rf $O/prof_ev
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ev -- $R/xlayer_prover_amd/host/zkprover_batch -c $R/tools/config_runFile_BatchProof_synthetic.json --proofs 1 > $O/prof_ev.out 2> $O/prof_ev.err
grep -E "evmap_partial|dot_columns" $(ls $O/prof_ev/*/*kernel_stats.csv | head -1) | cut -c1-40,150-260
python3 -c "
import json
j=json.loads(open('$O/prof_ev.out').read().strip().splitlines()[-1]); print(j['batch_proof_fnv'], j['timers_s']['zkevm.STARK_STEP_5_EVMAP'])"
