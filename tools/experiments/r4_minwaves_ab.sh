#!/bin/bash
# A/B: a floor on the waves per SIMD of the generated segment kernels (XLP_EXPR_MIN_WAVES at generation time -> __launch_bounds__(256, K)):
# plugins built into xlayer_prover_amd/chelpers/exp/mw{3,4} by
#   XLP_EXPR_MIN_WAVES=K python3 tools/build_expr_plugins.py --geometry 23 --aliased --opt-only --colmajor --steps step42ns,step3,step3prev --out-dir exp/mwK
# Prints the step-program stage timers of the second of two batch proofs for the product plugins and the two variants.
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4
for v in base mw3 mw4; do
  python3 - "$v" <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
from tools import make_stark_bundle as B
v = sys.argv[1]
pd = None if v == "base" else os.path.join("xlayer_prover_amd", "chelpers", "exp", v)
out = os.path.join("gpurun_out", "r4", "bundle_" + v)
B.make("zkevm", 23, True, out=out, colmajor=True, plugin_dir=os.path.abspath(pd) if pd else None)
cfg = json.load(open("tools/config_runFile_BatchProof_synthetic_colmajor.json"))
cfg["zkevmStarkInfo"] = os.path.join(out, "zkevm.starkinfo.json")
json.dump(cfg, open(os.path.join("gpurun_out", "r4", "cfg_" + v + ".json"), "w"))
PY
  timeout -k 10 300 xlayer_prover_amd/host/zkprover_batch -c gpurun_out/r4/cfg_$v.json --proofs 2 > gpurun_out/r4/minwaves_$v.txt 2> gpurun_out/r4/minwaves_$v.err
  python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
l = [json.loads(x) for x in open("gpurun_out/r4/minwaves_" + v + ".txt").read().strip().splitlines()][-1]
t = l["timers_s"]
print(v, {k.replace("zkevm.STARK_STEP_", ""): round(x, 4) for k, x in t.items() if "CALCULATE_EXPS" in k and "INTT" not in k and k.startswith("zkevm.")}, "zkevm", round(t["STARK_PROOF_BATCH_PROOF"], 3), l["batch_proof_fnv"])
PY
done
