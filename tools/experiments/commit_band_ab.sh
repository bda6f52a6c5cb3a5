# witness upload band width (XLP_COMMIT_BAND, words per row and band; `plan` = the library's own narrowing plan) against step 1 of the zkEVM STARK: second of two proofs each
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
for b in plan 64 256; do
  if [ $b = plan ]; then unset XLP_COMMIT_BAND; else export XLP_COMMIT_BAND=$b; fi
  timeout -k 10 300 xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_synthetic.json --proofs 2 > $O/band_$b.out 2> $O/band_$b.err || { tail -5 $O/band_$b.err; exit 1; }
  python3 - "$O/band_$b.out" $b <<'PY'
import json, sys
j = [json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{") and "timers_s" in l][-1]
print("band", sys.argv[2], "upload_ms", round(j["zkevm_witness_upload_ms"], 1), "step1_s", round(j["timers_s"]["zkevm.STARK_STEP_1_LDE_AND_MERKLETREE"], 4), "three_starks_s", round(j["three_starks_s"], 4), j["batch_proof_fnv"])
PY
done
