#!/bin/bash
# Round 5, seventh GPU call: register pressure of the step programs -- carry-free accumulators alive at once (a1 / a2; the product: 3), with the accumulator-worth rule
# (g: XLP_EXPR_LAZY_MIN_GAIN=190: no accumulator for a single product), other cuts / batch sizes on top of it, and reloads instead of long-held loaded words (r96).
set -e
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
bash tools/experiments/r5_variant_sweep.sh $R/variant_sweep_regs.txt base a2 a1 a2g a1g a2g_c256 a2g_c384 a2g_h2 a2g_r96 base
