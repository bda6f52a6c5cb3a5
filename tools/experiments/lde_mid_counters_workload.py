import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from xlayer_prover_amd import capi
capi.init(0); capi.use_torch_stream()
lg, ncols = 23, 8
n, n_ext = 1 << lg, 1 << (lg + 1)
src = torch.empty(n * ncols, dtype=torch.int64, device="cuda"); capi.dev_fill(src, 1, 0)
out = torch.empty(n_ext * ncols, dtype=torch.int64, device="cuda")
for m in ("0", "1"):
    os.environ["XLP_LDE_MFMA"] = m
    for _ in range(2): capi.dev_extend(out, src, n_ext, n, ncols)
capi.sync()
