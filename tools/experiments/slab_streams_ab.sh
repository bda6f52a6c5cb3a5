# step-program slabs: streams x workgroups per CU (XLP_EXPR_STREAMS, XLP_EXPR_SLAB_WGS) against the step stage timers: second of two proofs each
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
for cfg in "2 8" "3 8" "4 8" "2 12" "3 12" "2 6" "2 16"; do
  set -- $cfg
  XLP_EXPR_STREAMS=$1 XLP_EXPR_SLAB_WGS=$2 timeout -k 10 300 xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_synthetic.json --proofs 2 > $O/slab_$1_$2.out 2> $O/slab_$1_$2.err || { echo "streams $1 wgs $2: failed"; tail -3 $O/slab_$1_$2.err; continue; }
  python3 - "$O/slab_$1_$2.out" "$1" "$2" <<'PY'
import json, sys
j = [json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{") and "timers_s" in l][-1]
t = j["timers_s"]
k = ["zkevm.STARK_STEP_2_CALCULATE_EXPS", "zkevm.STARK_STEP_3_CALCULATE_EXPS", "zkevm.STARK_STEP_3_CALCULATE_EXPS_2", "zkevm.STARK_STEP_4_CALCULATE_EXPS_2NS"]
print("streams", sys.argv[2], "wgs/CU", sys.argv[3], [round(t[x], 4) for x in k], "sum", round(sum(t[x] for x in k), 4), "three_starks_s", round(j["three_starks_s"], 4), j["batch_proof_fnv"])
PY
done
