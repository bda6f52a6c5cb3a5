# extendPol scratch layout A/B on ONE box: libxlp_gl.so (8-column groups) against libxlp_gl_oldlayout.so ([row][chunk width]); the bench line
# (8 columns: both layouts are the same addresses there) and the extendPol kernels inside a batch proof (64-column chunks), interleaved
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
cd $R
cp xlayer_prover_amd/libxlp_gl.so /tmp/new.so; cp xlayer_prover_amd/libxlp_gl_oldlayout.so /tmp/old.so
for i in 1 2; do
  for v in new old; do
    cp /tmp/$v.so xlayer_prover_amd/libxlp_gl.so
    python bench.py --steps 20 --warmup 5 --no-genproof --no-cpu-baseline --no-bn128 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$v bench', j['value'], j['roofline']['avg_launch_ms'])"
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
done
cp /tmp/new.so xlayer_prover_amd/libxlp_gl.so
