"""A/B timing of the extendPol kernel selections (xlp_tune "lde_persist" bit mask: 1 = lde_top_dif, 2 = lde_mid, 4 = lde_top_dit as
persistent workgroups with LDS-direct prefetch) on BASELINE config 2 (2^23 x 8, blow-up 2) and on a 64-column chunk.
Per-kernel times are HIP events on the launch stream (xlp_profile_*).  Prints one JSON object."""
import json
import sys
import os
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlayer_prover_amd import capi


def run(lg, ncols, mask, steps=10):
    n, n_ext = 1 << lg, 1 << (lg + 1)
    src = torch.empty(n * ncols, dtype=torch.int64, device="cuda")
    out = torch.empty(n_ext * ncols, dtype=torch.int64, device="cuda")
    capi.dev_fill(src, 0x5EED0002, 0)
    capi.tune("lde_persist", mask)
    for _ in range(2):
        capi.dev_extend(out, src, n_ext, n, ncols)
    capi.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        capi.dev_extend(out, src, n_ext, n, ncols)
    capi.sync()
    wall = (time.perf_counter() - t0) / steps * 1e3
    capi.profile_enable(True)
    for _ in range(3):
        capi.dev_extend(out, src, n_ext, n, ncols)
    ms, cnt = capi.profile_read()
    capi.profile_enable(False)
    per = [m / 3 for m in ms]
    chk = int(out[::max(1, out.numel() // 4096)].sum().item()) & 0xFFFFFFFFFFFF
    return {"mask": mask, "ms_per_lde": round(wall, 4), "dif_mid_dit_ms": [round(x, 4) for x in per], "GFEs": round(n_ext * ncols / wall / 1e6, 2), "chk": chk}


def main():
    capi.init(0)
    capi.use_torch_stream()
    res = {}
    for lg, ncols in ((23, 8), (23, 64), (22, 8)):
        rows = [run(lg, ncols, m) for m in (0, 1, 2, 4, 7, 0, 7)]
        assert len({r["chk"] for r in rows}) == 1, "kernel selections disagree"
        res[f"2^{lg}x{ncols}"] = rows
    print(json.dumps(res))


if __name__ == "__main__":
    main()
