#!/usr/bin/env python3
"""Times Starks::evmap (starks.cpp:556-669) at zkEVM-like geometry: every column of a W-wide extended section
evaluated at xi (and a share of them at w*xi), n = 2^log_n rows of LEv/LpEv, section of 2^(log_n+1) rows."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlayer_prover_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=22)
ap.add_argument("--width", type=int, default=751)
ap.add_argument("--prime-share", type=float, default=0.25)
a = ap.parse_args()
capi.init(0); capi.use_torch_stream()
n, ne, W = 1 << a.log_n, 1 << (a.log_n + 1), a.width
sec = torch.empty(ne * W, dtype=torch.int64, device="cuda"); capi.dev_fill(sec, 1, 0)
LEv = torch.empty(3 * n, dtype=torch.int64, device="cuda"); capi.dev_fill(LEv, 2, 0)
LpEv = torch.empty(3 * n, dtype=torch.int64, device="cuda"); capi.dev_fill(LpEv, 3, 0)
ents = [(sec, W, c, 1, False) for c in range(W)] + [(sec, W, c, 1, True) for c in range(0, W, max(1, int(1 / a.prime_share)))]
ents = capi.evmap_entries(ents)
ev = torch.empty(3 * len(ents), dtype=torch.int64, device="cuda")
capi.dev_evmap(ev, LEv, LpEv, n, 1, ents); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    capi.dev_evmap(ev, LEv, LpEv, n, 1, ents)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
useful = len(ents) * n * 8
print(json.dumps({"log_n": a.log_n, "width": W, "evaluations": len(ents), "ms": round(ms, 3), "useful_GBps": round(useful / ms / 1e6, 1),
                  "rows_GBps": round(n * W * 8 / ms / 1e6, 1)}))
