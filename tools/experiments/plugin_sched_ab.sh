# A/B of the LLVM scheduling strategy for the ahead-of-time step-program kernels (round 3): default vs XLP_JIT_EXTRA_FLAGS="-mllvm -amdgpu-sched-strategy=<s>"
# (variants built beside the product plugins as *.so.<s>).  Runs on the GPU box.
C=xlayer_prover_amd/chelpers
run() { xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_synthetic.json --proofs 2 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); t=j['timers_s']; print('$1', round(j['three_starks_s'],4), {k.split('.')[1][11:]: round(v,4) for k,v in t.items() if 'CALCULATE_EXPS' in k and k.startswith('zkevm.') and 'INTT' not in k}, j['batch_proof_fnv'])"; }
for f in step2prev step3prev step3 step42ns; do cp $C/zkevm.${f}_opt.geo23a.so /tmp/$f.def; done
run default
for s in "$@"; do
  for f in step2prev step3prev step3 step42ns; do cp $C/zkevm.${f}_opt.geo23a.so.$s $C/zkevm.${f}_opt.geo23a.so; done
  run $s
  for f in step2prev step3prev step3 step42ns; do cp /tmp/$f.def $C/zkevm.${f}_opt.geo23a.so; done
done
run default
