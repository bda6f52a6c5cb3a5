#!/bin/bash
# Round 5, third GPU call: (1) Poseidon partial-round constants by scalar loads, A/B/A/B on ONE box (the first A/B said +5.3 %; on the next box the adopted kernel ran at
# the OLD kernel's speed of the first box: box-to-box spread or a fluke?) -- merkle leg only, product library vs gpurun_ab/poseidon_r4 (csrc/poseidon.hip of round 4);
# (2) the step-program variants generated with lazy canonicalisation + canonical sections (n*: XLP_EXPR_CANONICAL, csrc/expr.hip REPRESENTATIONS).
set -e
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
for i in 1 2; do
  python3 bench.py --no-e2e --no-genproof --no-bn128 --no-cpu-baseline --steps 5 > $R/poseidon_aba_new_$i.json 2>> $R/third.err
  XLP_LIB=$PWD/gpurun_ab/poseidon_r4/libxlp_gl.so python3 bench.py --no-e2e --no-genproof --no-bn128 --no-cpu-baseline --steps 5 > $R/poseidon_aba_r4_$i.json 2>> $R/third.err
done
python3 - <<'PY'
import json
for f in ("new_1", "r4_1", "new_2", "r4_2"):
    l = [json.loads(x) for x in open("gpurun_out/r5/poseidon_aba_" + f + ".json") if x.startswith("{")][-1]
    m = l["roofline_merkle"]
    print(f, "sponge ms", m["avg_launch_ms"], "row-major ms", m["row_major_sections"]["avg_launch_ms"], "resident ms", m["alu_ceiling"]["avg_launch_ms"], "Gperm/s", round(m["permutations_per_s"] / 1e9, 3))
PY
echo "poseidon aba done"
bash tools/experiments/r5_variant_sweep.sh $R/variant_sweep3.txt cut384 n384 n384h4 n384nolazy n320h4 n256h4 n512h4
