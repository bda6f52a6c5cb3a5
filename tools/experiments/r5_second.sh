#!/bin/bash
# Round 5, second GPU call: (1) parity of what changed in the library (Poseidon constants by scalar loads, the transposes' flat grid, the LDE kernels' resident
# twins next to unchanged product kernels), (2) the bench line with roofline.alu_ceiling / roofline_lde_colmajor, (3) the mid-pass kernel of extendPol with two
# workgroups per CU (gpurun_ab/lde_mid_v2: 2 columns x 512 threads per workgroup, 70 KB of LDS) against the product, (4) the step-program variants.
set -e
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
python3 -m pytest tests/test_gpu_parity.py tests/test_colmajor_kernels.py -q -m gpu -x -k "lde or extend or ntt or poseidon or merkle or colmajor or transpose or cm" --durations=10 > $R/second_parity.txt 2>&1 || { tail -30 $R/second_parity.txt; exit 1; }
tail -3 $R/second_parity.txt
python3 bench.py --no-e2e --no-genproof --no-bn128 --no-cpu-baseline > $R/lde_ab_a_product.json 2> $R/second.err
XLP_LIB=$PWD/gpurun_ab/lde_mid_v2/libxlp_gl.so python3 bench.py --no-e2e --no-genproof --no-bn128 --no-cpu-baseline > $R/lde_ab_b_mid_v2.json 2>> $R/second.err
XLP_LIB=$PWD/gpurun_ab/lde_mid_v2/libxlp_gl.so python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "extend or lde" > $R/lde_ab_b_parity.txt 2>&1 || { tail -30 $R/lde_ab_b_parity.txt; exit 1; }
python3 - <<'PY'
import json
for f in ("lde_ab_a_product", "lde_ab_b_mid_v2"):
    l = [json.loads(x) for x in open("gpurun_out/r5/" + f + ".json") if x.startswith("{")][-1]
    print(f, l["value"], l["roofline"]["avg_launch_ms"], "alu_ceiling", l["roofline"].get("alu_ceiling"))
    print(" cm", {k: v.get("GFE_s") for k, v in l.get("roofline_lde_colmajor", {}).items() if isinstance(v, dict) and "GFE_s" in v}, l.get("roofline_lde_colmajor", {}).get("error"))
PY
echo "lde ab done"
bash tools/experiments/r5_variant_sweep.sh $R/variant_sweep.txt base g8 g4 g2 seg384 seg512 cut384 cut512 cut768 cut512h12 cut512h4 cut768h4 cut1024h4 cut512h4a2 cut512h2a3
