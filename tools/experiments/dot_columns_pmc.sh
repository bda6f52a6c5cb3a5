# dot_columns alone (GPU box): timing, then PMC passes on the microbenchmark
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 300 python3 tools/experiments/bench_dot_columns.py --log-n 19 --reps 5 > $O/dc_bench.json 2> $O/dc_bench.err || { tail -20 $O/dc_bench.err; exit 1; }
cat $O/dc_bench.json
rocprofv3 -L > $O/rocprof_counters_list.txt 2>&1 || true
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf $O/pmc_dc_$tag
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_dc_$tag -- python3 tools/experiments/bench_dot_columns.py --log-n 19 --reps 1 > $O/pmc_dc_$tag.out 2> $O/pmc_dc_$tag.err || { tail -5 $O/pmc_dc_$tag.err; continue; }
  python3 - "$O/pmc_dc_$tag" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "dot_columns_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
print({k: (v / n[k]) for k, v in acc.items()}, dict(n))
PY
done
