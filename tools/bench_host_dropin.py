#!/usr/bin/env python3
"""The drop-in call pair of starks.cpp:53-56 on HOST pointers at the zkEVM size: xlp_extend(cm1_2ns, cm1_n) then
xlp_merkle(nodes, cm1_2ns), section 2^23 x 751 (50.4 GB in, 100.8 GB out, page-locked), with the device-copy cache off and on.
    python tools/bench_host_dropin.py [--log-n 23] [--width 751]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from xlayer_prover_amd import capi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=23)
    ap.add_argument("--width", type=int, default=751)
    a = ap.parse_args()
    capi.init(0)
    n, w = 1 << a.log_n, a.width
    n_ext = 2 * n
    h_in = capi.HostBuffer(n * w); h_out = capi.HostBuffer(n_ext * w); h_nodes = capi.HostBuffer((2 * n_ext - 1) * 4)
    tmp = torch.empty(n * w, dtype=torch.int64, device="cuda")
    capi.dev_fill(tmp, 0x5EED0401, 0); capi.copy_d2h(h_in, tmp)
    del tmp
    torch.cuda.empty_cache()
    lib = capi.load()
    res = {"workload": f"xlp_extend + xlp_merkle on host pointers, 2^{a.log_n} x {w} -> 2^{a.log_n + 1} x {w} (page-locked slab)",
           "GB": {"in": round(n * w * 8 / 1e9, 1), "out": round(n_ext * w * 8 / 1e9, 1), "nodes": round((2 * n_ext - 1) * 32 / 1e9, 2)}}
    roots = []
    for cache in (0, 200 << 30):
        capi.cache_enable(cache)
        best = None
        for rep in range(2):
            capi.cache_invalidate()
            t0 = time.perf_counter()
            capi._ck(lib.xlp_extend(h_out.ptr, h_in.ptr, n_ext, n, w, None), "xlp_extend")
            t1 = time.perf_counter()
            capi._ck(lib.xlp_merkle(h_nodes.ptr, h_out.ptr, w, n_ext), "xlp_merkle")
            t2 = time.perf_counter()
            cur = (t2 - t0, t1 - t0, t2 - t1)
            best = cur if best is None or cur[0] < best[0] else best
        roots.append([int(v) for v in h_nodes.array[-4:]])
        res["cache_on" if cache else "cache_off"] = {"total_s": round(best[0], 3), "xlp_extend_s": round(best[1], 3), "xlp_merkle_s": round(best[2], 3),
                                                    "pcie_GB_moved": round((n * w * 8 + n_ext * w * 8 * (1 if cache else 2) + (2 * n_ext - 1) * 32) / 1e9, 1)}
    capi.cache_enable(0)
    assert roots[0] == roots[1]
    res["root"] = roots[0]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
