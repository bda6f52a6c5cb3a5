#!/bin/bash
# round 4: band plans of the pipelined witness upload of step 1 on column-major sections (751 columns)
mkdir -p gpurun_out/r4
out=gpurun_out/r4/commit_band_plan_ab.txt
: > $out
run() {
  env "$@" timeout -k 10 150 xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_synthetic_colmajor.json --proofs 2 --timers > /tmp/ab.json 2> /tmp/ab.err
  python3 - "$*" >> $out <<'PY'
import json,sys
l=[json.loads(x) for x in open('/tmp/ab.json').read().strip().splitlines()]
t=l[-1]["timers_s"]
print(f"{sys.argv[1]:70s} three_starks {l[-1]['three_starks_s']:.4f}  step1 {t['zkevm.STARK_STEP_1_LDE_AND_MERKLETREE']:.4f}  upload_busy_ms {l[-1]['zkevm_witness_upload_ms']:.0f}  fnv {l[-1]['batch_proof_fnv']}")
PY
  tail -1 $out
}
run XLP_COMMIT_PLAN=0
run XLP_COMMIT_PLAN=1
run XLP_COMMIT_PLAN_WIDTHS=64,160,128,112,96,80,64,47
run XLP_COMMIT_PLAN_WIDTHS=96,160,128,112,96,80,79
run XLP_COMMIT_PLAN_WIDTHS=64,128,128,128,112,96,95
run XLP_COMMIT_PLAN_WIDTHS=32,96,160,160,128,96,79
run XLP_COMMIT_PLAN_WIDTHS=128,128,128,128,96,80,63
run XLP_COMMIT_PLAN=0
