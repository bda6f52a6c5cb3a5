#!/usr/bin/env python3
"""Times xlp_dev_merkle (leaf sponge + tree) for a few widths; prints perms/s and GB/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlayer_prover_amd import capi
capi.init(0); capi.use_torch_stream()
for (lg, w) in [(20, 8), (20, 64), (20, 751), (24, 8), (22, 168)]:
    h = 1 << lg
    src = torch.empty(h * w, dtype=torch.int64, device="cuda")
    capi.dev_fill(src, 0x5EED0003, 0)
    nodes = torch.empty((2 * h - 1) * 4, dtype=torch.int64, device="cuda")
    capi.dev_merkle(nodes, src, w, h); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps): capi.dev_merkle(nodes, src, w, h)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    perms = h * ((w + 7) // 8) + h - 1
    byts = 8 * h * w + 32 * (2 * h - 1)
    print(f"merkle 2^{lg} x {w}: {ms:.3f} ms  {perms/ms/1e6:.2f} Gperm/s  {byts/ms/1e6:.1f} GB/s", flush=True)
    del src, nodes
