# A/B of the LLVM scheduling strategy for the extendPol kernels (round 3): default vs -mllvm -amdgpu-sched-strategy=max-ilp (fewer s_nop, more VGPRs).
# Build the variant library first (see profiles/r3/lde_sched_strategy_ab.txt for the command); runs on the GPU box, interleaved.
for i in 1 2 3; do
  for lib in libxlp_gl.so libxlp_gl_ilp.so; do
    XLP_LIB=$PWD/xlayer_prover_amd/$lib python bench.py --steps 50 --warmup 10 --no-genproof --no-cpu-baseline --no-bn128 | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$lib', j['value'], j['roofline']['avg_launch_ms'])"
  done
done
