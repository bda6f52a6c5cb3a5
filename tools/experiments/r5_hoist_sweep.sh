#!/bin/bash
# NEXT ROUND (prepared at the end of round 4, not run): the batch size of the load pipelining of xlayer_prover_amd/expr_hoist.py per program.
# Round 4 measured ONE point (batch 8: step42ns 0.378 -> 0.301 s, step3 0.097 -> 0.089 s, step3prev 0.089 -> 0.091 s with a 436-register segment).
# Build the variants HERE (CPU, ~1 min each) into xlayer_prover_amd/chelpers/exp/hoist{0,4,6,12,16} by
#   for b in 0 4 6 12 16; do python3 tools/build_expr_plugins.py --geometry 23 --aliased --opt-only --colmajor --hoist-loads $b --out-dir xlayer_prover_amd/chelpers/exp/hoist$b; done
# then run this on the GPU box: it prints the step-program stage timers of the second of two batch proofs per variant and the proof's FNV (must be equal).
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
for v in base hoist0 hoist4 hoist6 hoist12 hoist16; do
  [ "$v" = base ] || [ -d xlayer_prover_amd/chelpers/exp/$v ] || continue
  python3 - "$v" <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
from tools import make_stark_bundle as B
v = sys.argv[1]
pd = None if v == "base" else os.path.join("xlayer_prover_amd", "chelpers", "exp", v)
out = os.path.join("gpurun_out", "r5", "bundle_" + v)
B.make("zkevm", 23, True, out=out, colmajor=True, plugin_dir=os.path.abspath(pd) if pd else None)
cfg = json.load(open("tools/config_runFile_BatchProof_synthetic_colmajor.json"))
cfg["zkevmStarkInfo"] = os.path.join(out, "zkevm.starkinfo.json")
json.dump(cfg, open(os.path.join("gpurun_out", "r5", "cfg_" + v + ".json"), "w"))
PY
  timeout -k 10 300 xlayer_prover_amd/host/zkprover_batch -c gpurun_out/r5/cfg_$v.json --proofs 2 > gpurun_out/r5/hoist_$v.txt 2> gpurun_out/r5/hoist_$v.err
  python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
l = [json.loads(x) for x in open("gpurun_out/r5/hoist_" + v + ".txt").read().strip().splitlines()][-1]
t = l["timers_s"]
print(v, {k.replace("zkevm.STARK_STEP_", ""): round(x, 4) for k, x in t.items() if "CALCULATE_EXPS" in k and "INTT" not in k and k.startswith("zkevm.")}, "three starks", round(l["three_starks_s"], 3), l["batch_proof_fnv"])
PY
done
