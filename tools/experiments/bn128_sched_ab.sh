# A/B of scheduling strategies for the BN128 kernels (bn128.hip, bn254_msm.hip): variant libraries, tools/bench_merkle_bn128.py + tools/bench_bn254.py
for lib in libxlp_gl.so libxlp_gl_bn_max-ilp.so libxlp_gl_bn_iterative-maxocc.so libxlp_gl.so; do
  echo $lib
  XLP_LIB=$PWD/xlayer_prover_amd/$lib python tools/bench_merkle_bn128.py 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('  merkle', [(r['leaves'], r['seconds']) for r in j['runs']])"
  XLP_LIB=$PWD/xlayer_prover_amd/$lib python tools/bench_bn254.py 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('  fft', [(r['n'], r['seconds']) for r in j['fft']], 'msm', [(r['n'], r['seconds']) for r in j['msm']])"
done
