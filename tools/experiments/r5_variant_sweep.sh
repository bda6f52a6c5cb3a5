#!/bin/bash
# Round 5: step-program plugin variants (xlayer_prover_amd/chelpers/exp/<name>, built on the CPU by tools/build_expr_plugins.py --out-dir with the
# generator options under test: --hoist-loads / --seg / --cut, XLP_EXPR_HOIST_AHEAD) through two batch proofs each on the GPU box: the step-program
# stage timers of the second proof and the proof's FNV (must be equal everywhere).   usage: r5_variant_sweep.sh <out.txt> base <name> <name> ...
set -e
cd "$(dirname "$0")/../.."
OUT=$1; shift
mkdir -p gpurun_out/r5
: > $OUT
for v in "$@"; do
  [ "$v" = base ] || [ -d xlayer_prover_amd/chelpers/exp/$v ] || { echo "$v: not built" >> $OUT; continue; }
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
  timeout -k 10 300 xlayer_prover_amd/host/zkprover_batch -c gpurun_out/r5/cfg_$v.json --proofs 2 > gpurun_out/r5/sweep_$v.txt 2> gpurun_out/r5/sweep_$v.err
  python3 - "$v" >> $OUT <<'PY'
import json, sys
v = sys.argv[1]
l = [json.loads(x) for x in open("gpurun_out/r5/sweep_" + v + ".txt").read().strip().splitlines()][-1]
t = l["timers_s"]
st = {k.replace("zkevm.STARK_STEP_", ""): round(x, 4) for k, x in t.items() if "CALCULATE_EXPS" in k and "INTT" not in k and k.startswith("zkevm.")}
print(v, st, "sum", round(sum(st.values()), 4), "three starks", round(l["three_starks_s"], 3), l["batch_proof_fnv"])
PY
  rm -rf gpurun_out/r5/bundle_$v
  tail -1 $OUT
done
