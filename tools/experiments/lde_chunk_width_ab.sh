# extendPol chunk width inside a proof (columns taken through the three passes together): 64 (HEAD) against 32 and 16 -- variant libraries, one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
cd $R
cp xlayer_prover_amd/libxlp_gl.so /tmp/w64.so; cp xlayer_prover_amd/libxlp_gl_w32.so /tmp/w32.so; cp xlayer_prover_amd/libxlp_gl_w16.so /tmp/w16.so
for v in w64 w32 w16 w64; do
  cp /tmp/$v.so xlayer_prover_amd/libxlp_gl.so
  rm -rf $O/prof_lde
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lde -- $R/xlayer_prover_amd/host/zkprover_batch -c $R/tools/config_runFile_BatchProof_synthetic.json --proofs 1 > $O/prof_lde.out 2> $O/prof_lde.err
  python3 - "$(ls $O/prof_lde/*/*kernel_stats.csv | head -1)" "$O/prof_lde.out" $v <<'PY'
import csv, sys, json
t = {}
for r in csv.DictReader(open(sys.argv[1])):
    for k in ("lde_mid", "lde_top_dit<11>", "lde_top_dif<11"):
        if k in r["Name"]: t[k] = round(float(r["TotalDurationNs"]) / 1e6, 1)
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[3], "in-proof ms", t, "sum", round(sum(t.values()), 1), j["batch_proof_fnv"])
PY
done
cp /tmp/w64.so xlayer_prover_amd/libxlp_gl.so
