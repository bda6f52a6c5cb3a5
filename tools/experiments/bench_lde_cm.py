#!/usr/bin/env python3
"""extendPol 2^23 x W, blow-up 2: the row-major three-kernel path against the column-major one (ntt_lde_cm.hip), per kernel (HIP events inside the
library: xlp_profile_*), same box, interleaved.  Prints one JSON line."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xlayer_prover_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=23)
ap.add_argument("--cols", default="8,64,168")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
capi.init(0)
n, ne = 1 << a.log_n, 2 << a.log_n
res = {}
for w in [int(x) for x in a.cols.split(",")]:
    src = torch.empty(n * w, dtype=torch.int64, device="cuda"); capi.dev_fill(src, 0x5EED0002, 0)
    out = torch.empty(ne * w, dtype=torch.int64, device="cuda")
    r = {}
    for name, fn in (("row_major", lambda: capi.dev_extend(out, src, ne, n, w)), ("col_major", lambda: capi.dev_extend_cm(out, ne, src, n, ne, n, w))):
        fn(); fn(); torch.cuda.synchronize()
        capi.profile_enable(True)
        for _ in range(a.reps):
            fn()
        ms, cnt = capi.profile_read()
        capi.profile_enable(False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        tot = e0.elapsed_time(e1) / a.reps
        r[name] = {"ms_per_lde": round(tot, 4), "GFE_s": round(ne * w / tot / 1e6, 2), "kernel_ms_per_lde": [round(x / a.reps, 4) for x in ms], "chunks": cnt // a.reps}
    res[f"2^{a.log_n}x{w}"] = r
    del src, out
print(json.dumps(res))
