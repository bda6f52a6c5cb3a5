#!/usr/bin/env python3
"""Runs ON THE GPU BOX, from the repo root.  HBM bytes and VALU instructions of the kernel classes of one Prover::genBatchProof
(host/zkprover_batch, one proof), each counter group in its own rocprofv3 --pmc pass, summed over all dispatches of a class:
    python3 tools/collect_step_counters.py gpurun_out/r2/step_counters.json [profiles/r2/genbatchproof_cpp_kernel_stats.csv]
With the kernel-stats CSV of a --kernel-trace run of the same command the totals become rates (TB/s of HBM traffic, wave-instructions/s).
FETCH_SIZE x 2 and WRITE_SIZE x 1 (KiB) as in tools/collect_counters.py / MI355X_MICROARCH.md."""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_WAVES"]]
CLASSES = [("seg_", "step-program segments (ahead-of-time)"), ("expr_gather", "column staging: gather"), ("expr_scatter", "column staging: scatter"),
           ("dot_columns", "dot_columns"), ("expr_kernel", "step-program interpreter"), ("linear_hash_absorb", "leaf sponge (banded)"),
           ("linear_hash_kernel", "leaf sponge"), ("merkle_level", "tree levels"),
           ("lde_cm_top_dif", "lde_cm_top_dif"), ("lde_cm_mid", "lde_cm_mid"), ("lde_cm_top_dit", "lde_cm_top_dit"),
           ("lde_top_dif", "lde_top_dif"), ("lde_mid", "lde_mid"), ("lde_top_dit", "lde_top_dit"),
           ("rows_to_cm", "transposes (witness bands, quotient)"), ("cm_to_rows", "transposes (witness bands, quotient)"), ("copy_view", "view copies (H1H2 / Z operands)"),
           ("copy_cols", "copy_cols"), ("scan3", "calculateZ scans"), ("h1h2", "calculateH1H2"), ("evmap", "evmap")]
CONFIG = os.environ.get("XLP_COUNTERS_CONFIG", os.path.join("tools", "config_runFile_BatchProof_synthetic_colmajor.json"))   # the product path: column-major sections


def cls(name):
    for key, label in CLASSES:
        if key in name:
            return label
    return None


def run_pass(counters):
    d = tempfile.mkdtemp(prefix="xlp_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", os.path.join(ROOT, "xlayer_prover_amd", "host", "zkprover_batch"),
           "-c", CONFIG, "--proofs", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise SystemExit(f"rocprofv3 failed: {r.stderr[-2000:]}")
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = cls(row["Kernel_Name"])
            if k is None:
                continue
            a = acc.setdefault(k, {})
            a[row["Counter_Name"]] = a.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return acc


def main():
    out_path = sys.argv[1]
    per = {}
    for p in PASSES:
        for k, cs in run_pass(p).items():
            per.setdefault(k, {}).update(cs)
        print("pass done:", p, flush=True)
    dur = {}
    if len(sys.argv) > 2:
        for row in csv.DictReader(open(sys.argv[2])):
            k = cls(row["Name"])
            if k:
                dur[k] = dur.get(k, 0.0) + float(row["TotalDurationNs"]) * 1e-9
    res = {}
    for k, cs in per.items():
        e = {"hbm_read_GB": round(cs.get("FETCH_SIZE", 0) * 2 * 1024 / 1e9, 2), "hbm_written_GB": round(cs.get("WRITE_SIZE", 0) * 1024 / 1e9, 2),
             "valu_wave_instructions": cs.get("SQ_INSTS_VALU"), "waves": cs.get("SQ_WAVES")}
        if k in dur:
            e["seconds_in_kernel_trace_run"] = round(dur[k], 4)
            e["hbm_TBps"] = round((e["hbm_read_GB"] + e["hbm_written_GB"]) / 1e3 / dur[k], 2)
            if e["valu_wave_instructions"]:
                e["valu_G_wave_instr_per_s"] = round(e["valu_wave_instructions"] / dur[k] / 1e9, 1)
        res[k] = e
    json.dump({"config": CONFIG, "source": "rocprofv3 --pmc <group> -- host/zkprover_batch --proofs 1 (the three STARKs of genBatchProof), summed per kernel class; "
                         "FETCH_SIZE x 2, WRITE_SIZE x 1, KiB", "per_class": res}, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
