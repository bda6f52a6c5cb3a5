#!/bin/bash
# Round 5, thirteenth GPU call: temporaries stored where they are last defined instead of at the end of their segment (XLP_EXPR_EARLY_STORES; es1) against the product
set -e
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
bash tools/experiments/r5_variant_sweep.sh $R/variant_sweep_early_stores.txt base es1 base es1
