#!/bin/bash
# Round 5, eighth GPU call: around ONE accumulator alive (a1g): no lazy accumulation at all (a0), batch sizes and cuts on top of a1g.
set -e
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
bash tools/experiments/r5_variant_sweep.sh $R/variant_sweep_regs2.txt base a1g a0 a1g_h8 a1g_c384 a1g_c512 a1g_h6c256 a1g
