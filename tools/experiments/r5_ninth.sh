#!/bin/bash
# Round 5, ninth GPU call: per-program generator options (tuned: step2prev 4,512 / step3prev 6,256 / step3 4,384 / step42ns 4,512, one accumulator alive, the
# accumulator-worth rule) with the FRI polynomial's 12-instruction row program compiled too (step52ns_drow: it ran interpreted, 27 ms per zkEVM proof), and probes
# around it (t_*: one program changed against `tuned`).  a1g_drow = a1g + the compiled row program alone.
set -e
cd "$(dirname "$0")/../.."
R=gpurun_out/r5
mkdir -p $R
bash tools/experiments/r5_variant_sweep.sh $R/variant_sweep_tuned.txt base a1g a1g_drow tuned t_42c640 t_42c768 t_42c512h8 t_3pc192h6 t_3pc256h8 t_3c384h8 tuned
