#!/bin/bash
# Round 5, sixth GPU call: run-time knobs of the step programs under the new generator (slabs in flight, slab size: no rebuild), and the number of carry-free
# accumulators alive at once in step42ns (XLP_EXPR_LAZY_ACCS 2 / 4 / 6 / 8; the product: 3).  Same batch proof everywhere.
set -e
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
: > $R/runtime_knobs.txt
run() {   # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 300 xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_synthetic_colmajor.json --proofs 2 > $R/knob_$label.txt 2> $R/knob_$label.err
  python3 - "$label" >> $R/runtime_knobs.txt <<'PY'
import json, sys
v = sys.argv[1]
l = [json.loads(x) for x in open("gpurun_out/r5/knob_" + v + ".txt").read().strip().splitlines()][-1]
t = l["timers_s"]
st = {k.replace("zkevm.STARK_STEP_", ""): round(x, 4) for k, x in t.items() if "CALCULATE_EXPS" in k and "INTT" not in k and k.startswith("zkevm.")}
print(v, st, "sum", round(sum(st.values()), 4), "three starks", round(l["three_starks_s"], 3), l["batch_proof_fnv"])
PY
  tail -1 $R/runtime_knobs.txt
}
run product XLP_NOOP=1
run streams1 XLP_EXPR_STREAMS=1
run streams3 XLP_EXPR_STREAMS=3
run wgs4 XLP_EXPR_SLAB_WGS=4
run wgs16 XLP_EXPR_SLAB_WGS=16
run wgs16s3 XLP_EXPR_SLAB_WGS=16 XLP_EXPR_STREAMS=3
run product_again XLP_NOOP=1
bash tools/experiments/r5_variant_sweep.sh $R/variant_sweep_accs.txt base accs2 accs4 accs6 accs8
