#!/usr/bin/env python3
"""Workload for PMC passes over the BN128 kernels: Poseidon-BN128 permutations of 2^18 states at t = 17 and 2^20 states at t = 5
(the two shapes of the recursivef trees), FFT 2^22 and an MSM of 2^18 points."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlayer_prover_amd import capi

capi.init(0)
for t, cnt in ((17, 1 << 18), (5, 1 << 20)):
    st = torch.empty(cnt * t * 4, dtype=torch.int64, device="cuda")
    capi.dev_fill(st, 0x5EED0F07, 0)
    st.view(-1, 4)[:, 3] &= 0x0FFFFFFFFFFFFFFF
    capi._ck(capi.load().xlp_dev_poseidon_bn128(capi._dp(st), t, cnt), "perm")
    torch.cuda.synchronize()
    del st
n = 1 << 22
a = torch.empty(n * 4, dtype=torch.int64, device="cuda"); capi.dev_fill(a, 0x5EED0F05, 0)
a.view(n, 4)[:, 3] &= 0x0FFFFFFFFFFFFFFF
capi.dev_fft_bn254(a, n)
torch.cuda.synchronize()
