"""A/B of extendPol's middle pass: the vector kernel (lde_mid) against the matrix-pipe kernel (lde_mid_mfma, XLP_LDE_MFMA=1).
1. Parity: the two must produce the same extended matrix, bit for bit, over sizes 2^17 .. 2^23, ragged widths and blow-ups 1 .. 8.
2. Timing: per-kernel HIP events on the launch stream (xlp_profile_*) on BASELINE config 2 (2^23 x 8, blow-up 2) and on a 64-column chunk.
Prints one JSON object."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xlayer_prover_amd import capi


def extend(lg, e, ncols, mfma, src=None, steps=0):
    os.environ["XLP_LDE_MFMA"] = str(int(mfma))            # 0: vector kernels, otherwise the matrix pipe
    n, n_ext = 1 << lg, 1 << (lg + e)
    if src is None:
        src = torch.empty(n * ncols, dtype=torch.int64, device="cuda")
        capi.dev_fill(src, 0x5EED0002 + lg, 0)
    out = torch.empty(n_ext * ncols, dtype=torch.int64, device="cuda")
    capi.dev_extend(out, src, n_ext, n, ncols)
    capi.sync()
    res = None
    if steps:
        for _ in range(2):
            capi.dev_extend(out, src, n_ext, n, ncols)
        capi.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            capi.dev_extend(out, src, n_ext, n, ncols)
        capi.sync()
        wall = (time.perf_counter() - t0) / steps * 1e3
        capi.profile_enable(True)
        for _ in range(5):
            capi.dev_extend(out, src, n_ext, n, ncols)
        ms, cnt = capi.profile_read()
        capi.profile_enable(False)
        res = {"mfma": mfma, "ms_per_lde": round(wall, 4), "dif_mid_dit_ms": [round(x / 5, 4) for x in ms], "GFEs": round(n_ext * ncols / wall / 1e6, 2)}
    return out, src, res


def extend_cm(lg, e, ncols, mfma, src=None, steps=0):
    os.environ["XLP_LDE_MFMA"] = str(int(mfma))            # 0: vector kernels, otherwise the matrix pipe
    n, n_ext = 1 << lg, 1 << (lg + e)
    if src is None:
        src = torch.empty(n * ncols, dtype=torch.int64, device="cuda")
        capi.dev_fill(src, 0x5EED0003 + lg, 0)
    out = torch.empty(n_ext * ncols, dtype=torch.int64, device="cuda")
    capi.dev_extend_cm(out, n_ext, src, n, n_ext, n, ncols)
    capi.sync()
    res = None
    if steps:
        for _ in range(2):
            capi.dev_extend_cm(out, n_ext, src, n, n_ext, n, ncols)
        capi.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            capi.dev_extend_cm(out, n_ext, src, n, n_ext, n, ncols)
        capi.sync()
        wall = (time.perf_counter() - t0) / steps * 1e3
        capi.profile_enable(True)
        for _ in range(5):
            capi.dev_extend_cm(out, n_ext, src, n, n_ext, n, ncols)
        ms, cnt = capi.profile_read()
        capi.profile_enable(False)
        res = {"mfma": mfma, "layout": "column-major", "ms_per_lde": round(wall, 4), "dif_mid_dit_ms": [round(x / 5, 4) for x in ms], "GFEs": round(n_ext * ncols / wall / 1e6, 2)}
    return out, src, res


def main():
    capi.init(0)
    capi.use_torch_stream()
    report = {"parity": [], "timing": {}}
    cases = [(17, 1, 8), (17, 0, 4), (18, 2, 5), (19, 3, 7), (20, 1, 12), (21, 1, 9), (22, 1, 8), (23, 1, 8), (20, 1, 70)]
    quick = "--quick" in sys.argv
    if quick:
        cases = [(17, 1, 8), (19, 3, 7), (23, 1, 8)]
    ok = True
    for lg, e, ncols in cases:
        a, src, _ = extend(lg, e, ncols, 0)
        b, _, _ = extend(lg, e, ncols, 1, src=src)
        same = bool(torch.equal(a, b))
        nbad = 0 if same else int((a != b).sum().item())
        report["parity"].append({"lg": lg, "e": e, "ncols": ncols, "equal": same, "mismatches": nbad})
        ok = ok and same
        del a, b, src
    for lg, e, ncols in ([(17, 1, 8), (19, 3, 7), (23, 1, 8)] if quick else [(17, 1, 8), (17, 0, 4), (18, 2, 5), (19, 3, 7), (20, 1, 33), (22, 1, 8), (23, 1, 8)]):
        a, src, _ = extend_cm(lg, e, ncols, 0)
        b, _, _ = extend_cm(lg, e, ncols, 1, src=src)
        same = bool(torch.equal(a, b))
        report["parity"].append({"layout": "column-major", "lg": lg, "e": e, "ncols": ncols, "equal": same, "mismatches": 0 if same else int((a != b).sum().item())})
        ok = ok and same
        del a, b, src
    report["parity_ok"] = ok
    if "--time" in sys.argv or quick:
        for lg, ncols in (((23, 8),) if quick else ((23, 8), (23, 64), (22, 8))):
            rows = []
            for m in (0, 1, 0, 1):
                _, _, r = extend(lg, 1, ncols, m, steps=10)
                rows.append(r)
            report["timing"][f"2^{lg}x{ncols}"] = rows
            rows = []
            for m in (0, 1, 0, 1):
                _, _, r = extend_cm(lg, 1, ncols, m, steps=10)
                rows.append(r)
            report["timing"][f"2^{lg}x{ncols} column-major"] = rows
    print(json.dumps(report))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
