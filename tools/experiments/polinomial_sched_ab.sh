# A/B: csrc/polinomial.hip (dot_columns, calculateZ scans, evmap) with the default scheduler vs -mllvm -amdgpu-sched-strategy=max-ilp (variant library in xlayer_prover_amd/alt/)
run() { "$@" xlayer_prover_amd/host/zkprover_batch -c tools/config_runFile_BatchProof_synthetic.json --proofs 2 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); t=j['timers_s']; print(round(j['three_starks_s'],4), {k.split('.')[1][11:]: round(v,4) for k,v in t.items() if k.startswith('zkevm.') and any(x in k for x in ('STEP_5','CALCULATE_Z','H1H2'))}, j['batch_proof_fnv'])"; }
echo default; run env
echo max-ilp; run env LD_LIBRARY_PATH=$PWD/xlayer_prover_amd/alt
echo default; run env
echo max-ilp; run env LD_LIBRARY_PATH=$PWD/xlayer_prover_amd/alt
