#!/bin/bash
# round 4: slab size / stream count of the step programs on COLUMN-MAJOR sections (no staging copy any more: the slab is only the unit of the
# temporaries' scratch and of the launch tails).  zkprover_batch --proofs 2, second proof's stage timers.
mkdir -p gpurun_out/r4
out=gpurun_out/r4/steps_slab_ab.txt
: > $out
for cfg in "8 2" "4 2" "16 2" "32 2" "8 1" "16 1" "16 3" "64 2"; do
  set -- $cfg
  XLP_EXPR_SLAB_WGS=$1 XLP_EXPR_STREAMS=$2 timeout -k 10 150 xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_synthetic_colmajor.json --proofs 2 --timers > /tmp/ab.json 2> /tmp/ab.err
  python3 - "$1" "$2" >> $out <<'PY'
import json,sys
l=[json.loads(x) for x in open('/tmp/ab.json').read().strip().splitlines()]
t=l[-1]["timers_s"]
k=["STARK_STEP_2_CALCULATE_EXPS","STARK_STEP_3_CALCULATE_EXPS","STARK_STEP_3_CALCULATE_EXPS_2","STARK_STEP_4_CALCULATE_EXPS_2NS","STARK_STEP_5_CALCULATE_EXPS"]
print(f"SLAB_WGS={sys.argv[1]:>3} STREAMS={sys.argv[2]}  three_starks {l[-1]['three_starks_s']:.4f}  " + "  ".join(f"{x[11:]}={t['zkevm.'+x]:.4f}" for x in k) + f"  fnv {l[-1]['batch_proof_fnv']}")
PY
  tail -1 $out
done
