# stage timers of the step programs in one batch proof (GPU box): two proofs, the second one's timers
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 300 xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_synthetic.json --proofs 2 > $O/steps_timers.out 2> $O/steps_timers.err || { tail -20 $O/steps_timers.err; exit 1; }
grep -c compiled $O/steps_timers.err
python3 - "$O/steps_timers.out" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if not line.startswith("{"): continue
    j = json.loads(line)
    if "timers_s" not in j: print(line[:300]); continue
    t = j["timers_s"]
    keys = ["zkevm.STARK_STEP_2_CALCULATE_EXPS", "zkevm.STARK_STEP_3_CALCULATE_EXPS", "zkevm.STARK_STEP_3_CALCULATE_EXPS_2", "zkevm.STARK_STEP_4_CALCULATE_EXPS_2NS", "zkevm.STARK_STEP_5_CALCULATE_EXPS"]
    print(j["run"], j["batch_proof_fnv"], {k.split("STARK_")[1]: round(t[k], 4) for k in keys}, "sum", round(sum(t[k] for k in keys), 4), "three_starks_s", round(j["three_starks_s"], 4))
PY
