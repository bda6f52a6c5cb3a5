#!/usr/bin/env python3
"""xlp_dev_dot_columns alone, on the item list of the zkEVM FRI polynomial (tools/make_stark_bundle.py at 2^LOG_N rows, synthetic column words):
    python3 tools/experiments/bench_dot_columns.py [--log-n 19] [--reps 5]
prints ms per launch, the HBM rate of the column words read once, and the products per second.  Run it under rocprofv3 --pmc for counters."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=19)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    from tools import make_stark_bundle as B
    from xlayer_prover_amd import capi as xlp
    xlp.init()
    d = B.make("zkevm", a.log_n, False, out=tempfile.mkdtemp(prefix="xlp_bdc_"))
    si = json.load(open(os.path.join(d, "zkevm.starkinfo.json")))
    man = json.load(open(os.path.join(d, "zkevm.steps.json")))["steps"]["step52ns"]
    w = np.fromfile(os.path.join(d, man["dots"]), dtype=np.uint64)
    n, ngroups = int(w[2]), int(w[1])
    e = w[3:].reshape(n, 6)
    next_rows = 1 << (a.log_n + 1)
    pols = torch.randint(0, 2**62, (int(si["mapTotalN"]),), dtype=torch.int64, device="cuda")
    nconst = int(si["nConstants"])
    consts = torch.randint(0, 2**62, (nconst * next_rows,), dtype=torch.int64, device="cuda")
    coefs = torch.randint(0, 2**62, (3 * int(man["n_ext"]),), dtype=torch.int64, device="cuda")
    out = torch.zeros(next_rows * 3 * ngroups, dtype=torch.int64, device="cuda")
    arr = (xlp.DotEntry * n)()
    cols = set()
    for i, (kind, off, stride, dim, group, coef) in enumerate(e.tolist()):
        base = pols.data_ptr() if kind == 0 else consts.data_ptr()
        arr[i] = xlp.DotEntry(base + 8 * off, stride if kind == 0 else nconst, dim, group, coef)
        cols.add((kind, off))
    torch.cuda.synchronize()
    xlp.dev_dot_columns(out, arr, ngroups, coefs, next_rows); xlp.sync()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        xlp.dev_dot_columns(out, arr, ngroups, coefs, next_rows)
    xlp.sync()
    ms = (time.perf_counter() - t0) / a.reps * 1e3
    gb = len(cols) * next_rows * 8 / 1e9
    print(json.dumps({"rows": next_rows, "items": n, "columns": len(cols), "groups": ngroups, "ms_per_launch": round(ms, 3), "column_GB": round(gb, 2),
                      "TBps": round(gb / ms, 3), "G_base_products_per_s": round(3 * n * next_rows / ms / 1e6, 1), "checksum": int(out.sum().item())}))


if __name__ == "__main__":
    main()
