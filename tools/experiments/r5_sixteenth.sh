#!/bin/bash
# Round 5, sixteenth GPU call: the run-time knobs of the step programs once more, on the FINAL plugins (one accumulator, per-program options)
set -e
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
: > $R/runtime_knobs_final.txt
run() {
  local label=$1; shift
  env "$@" timeout -k 10 300 xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_synthetic_colmajor.json --proofs 2 > $R/knobf_$label.txt 2> $R/knobf_$label.err
  python3 - "$label" >> $R/runtime_knobs_final.txt <<'PY'
import json, sys
v = sys.argv[1]
l = [json.loads(x) for x in open("gpurun_out/r5/knobf_" + v + ".txt").read().strip().splitlines()][-1]
t = l["timers_s"]
st = {k.replace("zkevm.STARK_STEP_", ""): round(x, 4) for k, x in t.items() if "CALCULATE_EXPS" in k and "INTT" not in k and k.startswith("zkevm.")}
print(v, st, "sum", round(sum(st.values()), 4), "three starks", round(l["three_starks_s"], 3), l["batch_proof_fnv"])
PY
  tail -1 $R/runtime_knobs_final.txt
}
run product XLP_NOOP=1
run wgs4 XLP_EXPR_SLAB_WGS=4
run wgs6 XLP_EXPR_SLAB_WGS=6
run wgs12 XLP_EXPR_SLAB_WGS=12
run wgs16 XLP_EXPR_SLAB_WGS=16
run streams3 XLP_EXPR_STREAMS=3
run product_again XLP_NOOP=1
