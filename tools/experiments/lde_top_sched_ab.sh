# A/B of scheduling strategies for lde_top_dif / lde_top_dit (lde_mid keeps max-ilp): default vs max-memory-clause vs iterative-maxocc variant libraries
for i in 1 2; do
  for lib in libxlp_gl.so libxlp_gl_max-memory-clause.so libxlp_gl_iterative-maxocc.so; do
    XLP_LIB=$PWD/xlayer_prover_amd/$lib python bench.py --steps 50 --warmup 10 --no-genproof --no-cpu-baseline --no-bn128 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$lib', j['value'], j['roofline']['avg_launch_ms'])"
  done
done
