#!/usr/bin/env python3
"""Workload for the PMC passes of tools/collect_counters.py (rocprofv3 --pmc ..., one counter group per run):
  1. a calibration copy with a known byte count in the kernels' own access width (8 B per lane),
  2. the bench workload: extendPol 2^23 x 8 -> 2^24 x 8 (BASELINE configs[1]), 3 times,
  3. a Merkle commit 2^20 x 751 (the leaf sponge that is 36 % of a zkEVM proof), 2 times, row-major and on column-major sections (the product path since round 4),
  4. extendPol of the same 2^23 x 8 on column-major sections (ntt_lde_cm.hip), 3 times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlayer_prover_amd import capi

capi.init(0)
cnt = 1 << 27                                   # 1 GiB read + 1 GiB written by copy_canon_kernel
a = torch.empty(cnt, dtype=torch.int64, device="cuda"); b = torch.empty_like(a)
capi.dev_fill(a, 1, 0)
capi.dev_ntt(b, a, 1, cnt)                      # n = 1: the canonicalising copy kernel, 8 B/lane loads and stores
del a, b
n, ncols = 1 << 23, 8
src = torch.empty(n * ncols, dtype=torch.int64, device="cuda"); out = torch.empty(2 * n * ncols, dtype=torch.int64, device="cuda")
capi.dev_fill(src, 0x5EED0002, 0)
for _ in range(3):
    capi.dev_extend(out, src, 2 * n, n, ncols)
torch.cuda.synchronize()
del src, out
h, w = 1 << 20, 751
rows = torch.empty(h * w, dtype=torch.int64, device="cuda"); nodes = torch.empty((2 * h - 1) * 4, dtype=torch.int64, device="cuda")
capi.dev_fill(rows, 0x5EED0003, 0)
for _ in range(2):
    capi.dev_merkle(nodes, rows, w, h)
torch.cuda.synchronize()
# the same leaves from a column-major section (lane = row: 512 contiguous bytes per wave and word)
for _ in range(2):
    capi.dev_merkle_cm(nodes, rows, h, w, h, 0)
torch.cuda.synchronize()
del rows, nodes
src = torch.empty(n * ncols, dtype=torch.int64, device="cuda"); out = torch.empty(2 * n * ncols, dtype=torch.int64, device="cuda")
capi.dev_fill(src, 0x5EED0002, 0)
for _ in range(3):
    capi.dev_extend_cm(out, 2 * n, src, n, 2 * n, n, ncols)
torch.cuda.synchronize()
