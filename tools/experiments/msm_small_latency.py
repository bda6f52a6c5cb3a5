import torch, json, sys, time
sys.path.insert(0, ".")
import numpy as np
from xlayer_prover_amd import capi
from oracle import bn128_oracle as bo
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
capi.init(0); capi.use_torch_stream()
n = 4096
pts=[]; cur=None
for i in range(64):
    cur = bo.g1_add(cur, bo.G1); pts.append(cur)
tab = np.array([bo.g1_to_limbs(p) for p in pts], dtype=np.uint64)
bases = capi.from_np(tab[np.random.default_rng(1).integers(0, 64, size=n)].reshape(-1))
sc = torch.empty(n * 4, dtype=torch.int64, device=dev); capi.dev_fill(sc, 0x5EED0F06, 0)
res = torch.empty(8, dtype=torch.int64, device=dev)
for _ in range(3): capi.dev_msm_bn254(res, bases, sc, n)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(50): capi.dev_msm_bn254(res, bases, sc, n)
torch.cuda.synchronize()
print("ms per 4096-point msm", (time.perf_counter()-t)/50*1e3)
