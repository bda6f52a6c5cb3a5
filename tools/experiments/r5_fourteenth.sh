#!/bin/bash
# Round 5, fourteenth GPU call: gl::lazy3 without its three base registers (the base enters the accumulators as two halves; fold() one addc shorter; l3) against the product
set -e
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
bash tools/experiments/r5_variant_sweep.sh $R/variant_sweep_lazy3_base.txt base l3 base l3
