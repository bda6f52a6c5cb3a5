# A/B of the LLVM scheduling strategy for poseidon.hip (leaf sponge = 40 % of a batch proof): variant libraries libxlp_gl_pos_<strategy>.so built with
#   hipcc ... -mllvm -amdgpu-sched-strategy=<strategy> -c poseidon.hip   (other objects as in the product); interleaved runs on one GPU box
for i in 1 2; do
  for lib in libxlp_gl.so libxlp_gl_pos_max-ilp.so libxlp_gl_pos_iterative-maxocc.so libxlp_gl_pos_iterative-ilp.so libxlp_gl_pos_iterative-minreg.so libxlp_gl_pos_max-memory-clause.so; do
    XLP_LIB=$PWD/xlayer_prover_amd/$lib python bench.py --steps 5 --warmup 2 --no-genproof --no-cpu-baseline --no-bn128 | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); m=j['roofline_merkle']; print('$lib', {k: m[k] for k in m if 'perm' in k or 'ms' in k})"
  done
done
