#!/usr/bin/env python3
"""extendPol of a 64-column band: the band alone (row stride 64 words) against the same band inside a 751-wide and a 752-wide row-major
section (row strides 6 008 / 6 016 bytes) -- what the banded section commit of Starks::genProof runs.  Per-kernel times from the library's
HIP events.  Runs ON THE GPU BOX."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlayer_prover_amd import capi


def main():
    capi.init(0)
    ln = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    n, ne, wc = 1 << ln, 2 << ln, 64
    res = {}
    for ld, c0 in ((64, 0), (751, 64), (751, 320), (752, 64), (768, 64)):
        src = torch.empty(n * ld, dtype=torch.int64, device="cuda"); capi.dev_fill(src, 7, 0)
        out = torch.empty(ne * ld, dtype=torch.int64, device="cuda")
        for _ in range(2):
            capi.dev_extend_cols(out, src, ne, n, ld, c0, wc)
        torch.cuda.synchronize()
        capi.profile_enable(True)
        reps = 5
        for _ in range(reps):
            capi.dev_extend_cols(out, src, ne, n, ld, c0, wc)
        ms, cnt = capi.profile_read()
        capi.profile_enable(False)
        res[f"ld={ld} col0={c0}"] = {"kernel_ms": [round(x / cnt, 3) for x in ms], "sum_ms": round(sum(ms) / cnt, 3), "launches": cnt}
        del src, out
        torch.cuda.empty_cache()
    print(json.dumps({"workload": f"extendPol 2^{ln} -> 2^{ln + 1}, band of {wc} columns", "results": res}))


if __name__ == "__main__":
    main()
