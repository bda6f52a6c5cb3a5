# SQ_* counters of the vector and the matrix-pipe middle pass of extendPol on the same input (two rocprofv3 --pmc passes); output: profiles/r5/lde_mid_mfma_counters_vs_vector.txt
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/pmc_out; timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_out -- python3 $GRAFT_REPO_ROOT/tools/experiments/lde_mid_counters_workload.py > /tmp/pmc.log 2>&1 || tail -5 /tmp/pmc.log
  python3 - <<'PY'
import csv, glob
acc = {}
for f in glob.glob("/tmp/pmc_out/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "lde_mid" not in k: continue
        k = "mfma" if "mfma" in k else "valu"
        a = acc.setdefault(k, {}).setdefault(row["Counter_Name"], {})
        a[row["Dispatch_Id"]] = a.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
for k, cs in acc.items():
    print(k, {c: round(sum(v.values()) / len(v)) for c, v in cs.items()})
PY
done
