"""fflonk building blocks on one MI355X: FFT over the BN254 scalar field and G1 multi-scalar multiplication (device-resident inputs).
Prints one JSON line."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xlayer_prover_amd import capi


def main():
    capi.init(0); capi.use_torch_stream()
    res = {"fft": [], "msm": []}
    for lg in (20, 22, 24):
        n = 1 << lg
        a = torch.empty(n * 4, dtype=torch.int64, device="cuda")
        capi.dev_fill(a, 0x5EED0F05, 0)
        a.view(n, 4)[:, 3] &= 0x0FFFFFFFFFFFFFFF
        capi.dev_fft_bn254(a, n); capi.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            capi.dev_fft_bn254(a, n)
        capi.sync()
        dt = (time.perf_counter() - t0) / 3
        res["fft"].append({"n": f"2^{lg}", "seconds": round(dt, 5), "Melem_per_s": round(n / dt / 1e6, 1)})
    # MSM: 2^lg copies of a few curve points (multiples of the generator computed on the host with Python integers)
    q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    R = (1 << 256) % q

    def add(p, r):
        if p is None: return r
        if p == r:
            l = 3 * p[0] * p[0] * pow(2 * p[1], -1, q) % q
        else:
            l = (r[1] - p[1]) * pow(r[0] - p[0], -1, q) % q
        x = (l * l - p[0] - r[0]) % q
        return x, (l * (p[0] - x) - p[1]) % q
    pts, cur = [], None
    for _ in range(1024):
        cur = add(cur, (1, 2)); pts.append(cur)
    tab = np.array([[((c * R % q) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for c in p for i in range(4)] for p in pts], dtype=np.uint64)
    for lg in (16, 20, 22):
        n = 1 << lg
        rng = np.random.default_rng(lg)
        bases = capi.from_np(tab[rng.integers(0, 1024, size=n)].reshape(-1))
        sc = torch.empty(n * 4, dtype=torch.int64, device="cuda")
        capi.dev_fill(sc, 0x5EED0F06, 0)
        out = torch.empty(8, dtype=torch.int64, device="cuda")
        capi.dev_msm_bn254(out, bases, sc, n); capi.sync()
        t0 = time.perf_counter()
        for _ in range(2):
            capi.dev_msm_bn254(out, bases, sc, n)
        capi.sync()
        dt = (time.perf_counter() - t0) / 2
        res["msm"].append({"n": f"2^{lg}", "seconds": round(dt, 4), "Mpoints_per_s": round(n / dt / 1e6, 2)})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
