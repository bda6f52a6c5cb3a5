#!/usr/bin/env python3
"""Times stand-alone xlp_dev_ntt (generic pass engine) at the sizes Starks::genProof uses (starks.cpp:262,285,326)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlayer_prover_amd import capi
capi.init(0); capi.use_torch_stream()
for (lg, w, inv) in [(24, 3, True), (24, 6, False), (23, 3, True), (20, 1, False), (24, 8, False), (22, 64, False)]:
    n = 1 << lg
    src = torch.empty(n * w, dtype=torch.int64, device="cuda"); dst = torch.empty_like(src)
    capi.dev_fill(src, 0x5EED0001, 0)
    capi.dev_ntt(dst, src, n, w, inv); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps): capi.dev_ntt(dst, src, n, w, inv)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{'INTT' if inv else 'NTT '} 2^{lg} x {w}: {ms:.3f} ms  {n*w/ms/1e6:.2f} GFE/s  {16*n*w/ms/1e6:.1f} GB/s (16 B/FE)", flush=True)
