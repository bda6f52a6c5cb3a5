#!/usr/bin/env python3
"""zkEVM-scale timings of the row-scale helpers of genProof (polinomial.hpp): calculateH1H2, calculateZ,
batchInverse, xDivXSubXi, geometric tables."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xlayer_prover_amd import capi

capi.init(0); capi.use_torch_stream()
LOGN = int(sys.argv[1]) if len(sys.argv) > 1 else 23
n = 1 << LOGN


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def z(cnt): return torch.zeros(cnt, dtype=torch.int64, device="cuda")
def rnd(cnt, seed):
    t = torch.empty(cnt, dtype=torch.int64, device="cuda"); capi.dev_fill(t, seed, 0); return t


out = {}
for dim in (1, 3):
    # plookup shape: t = a table with repeats, f = values drawn from it (skewed)
    base = rnd(n * dim, 5).view(n, dim)
    idx_t = torch.sort(torch.randint(0, n // 4, (n,), device="cuda")).values
    t = base[idx_t].contiguous().view(-1)
    f = t.view(n, dim)[(torch.rand(n, device="cuda") ** 3 * n).long().clamp(max=n - 1)].contiguous().view(-1)
    h1, h2 = z(n * dim), z(n * dim)
    out[f"calculateH1H2 dim{dim}"] = timed(lambda: capi.dev_calculate_h1h2(h1, h2, f, t, n, dim))
num, den, zz = rnd(3 * n, 7), rnd(3 * n, 8), z(3 * n)
out["calculateZ"] = timed(lambda: capi.dev_calculate_z(zz, num, den, n))
ne = 2 * n
src, res = rnd(3 * ne, 9), z(3 * ne)
out["batchInverse ext (2N)"] = timed(lambda: capi.dev_batch_inverse(res, src, ne, 3))
x = rnd(ne, 10)
out["xDivXSubXi (2N)"] = timed(lambda: capi.dev_x_div_x_sub_xi(res, x, np.array([3, 5, 7], np.uint64), ne))
g = z(3 * n)
out["LEv table (ext, N)"] = timed(lambda: capi.dev_geometric(g, n, 3, np.array([1, 0, 0], np.uint64), np.array([3, 5, 7], np.uint64)))
g1 = z(ne)
out["x_2ns table (2N)"] = timed(lambda: capi.dev_geometric(g1, ne, 1, np.array([7], np.uint64), np.array([12345], np.uint64)))
print(json.dumps({"log_n": LOGN, "ms": {k: round(v, 3) for k, v in out.items()}}))
