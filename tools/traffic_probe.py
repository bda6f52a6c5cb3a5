#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs):
a calibration copy with a known byte count in the kernels' own access width (8 B per lane), then the
bench workload (LDE 2^23 x 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlayer_prover_amd import capi
capi.init(0); capi.use_torch_stream()
cnt = 1 << 27                                   # 1 GiB read + 1 GiB written by copy_canon_kernel
a = torch.empty(cnt, dtype=torch.int64, device="cuda"); b = torch.empty_like(a)
capi.dev_fill(a, 1, 0)
capi.dev_ntt(b, a, 1, cnt)                      # n = 1: the canonicalising copy kernel, 8 B/lane loads and stores
n, ncols = 1 << 23, 8
src = torch.empty(n * ncols, dtype=torch.int64, device="cuda"); out = torch.empty(2 * n * ncols, dtype=torch.int64, device="cuda")
capi.dev_fill(src, 0x5EED0002, 0)
for _ in range(3):
    capi.dev_extend(out, src, 2 * n, n, ncols)
torch.cuda.synchronize()
