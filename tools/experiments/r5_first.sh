#!/bin/bash
# Round 5, first GPU call: the tree as round 4 left it (bench line), then the three A/Bs round 4 prepared and never ran:
#   poseidon partial-round constants by scalar loads (gpurun_ab/poseidon_as4), the MSM bucket pass with the next point in flight (gpurun_ab/msm_prefetch),
#   the batch size of the load pipelining per program (xlayer_prover_amd/chelpers/exp/hoist{4,6,12,16}; r5_hoist_sweep.sh).
set -e
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
python3 bench.py > $R/bench_head.json 2> $R/bench_head.err
echo "bench head done"
# --- Poseidon A/B
python3 bench.py --no-e2e --no-genproof --no-bn128 --no-cpu-baseline > $R/poseidon_ab_a_product.json 2>> $R/bench_head.err
XLP_LIB=$PWD/gpurun_ab/poseidon_as4/libxlp_gl.so python3 bench.py --no-e2e --no-genproof --no-bn128 --no-cpu-baseline > $R/poseidon_ab_b_scalar_constants.json 2>> $R/bench_head.err
XLP_LIB=$PWD/gpurun_ab/poseidon_as4/libxlp_gl.so python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "poseidon or merkle" -x > $R/poseidon_ab_b_parity.txt 2>&1
echo "poseidon ab done"
# --- MSM A/B
xlayer_prover_amd/host/test_host_shim fflonk_bench 22 3 > $R/msm_ab_a_product.txt 2>&1
LD_LIBRARY_PATH=$PWD/gpurun_ab/msm_prefetch:$LD_LIBRARY_PATH xlayer_prover_amd/host/test_host_shim fflonk_bench 22 3 > $R/msm_ab_b_prefetch.txt 2>&1
XLP_LIB=$PWD/gpurun_ab/msm_prefetch/libxlp_gl.so python3 -m pytest tests/test_bn128.py -q -m gpu -k msm -x > $R/msm_ab_b_parity.txt 2>&1
echo "msm ab done"
# --- hoist batch sweep (the round-4 text pass)
bash tools/experiments/r5_hoist_sweep.sh > $R/hoist_sweep.txt 2>&1
echo "hoist sweep done"
