#!/usr/bin/env python3
"""Stage timings of the complete toy prover (tests/mini_genproof_flow.py: every stage of Starks::genProof on a
small PIL) on one MI355X, GPU-resident.  The trace is narrow (2 + 6 + 3 + 6 columns), so this shows the stage
structure and the fixed costs, not zkEVM-scale throughput (tools/bench_genproof.py, tools/bench_expr.py do that).
    python tools/bench_mini_genproof.py [--nbits 22] [--queries 64]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import mini_genproof_flow as flow
from tests.mini_backends import GpuBackend
from xlayer_prover_amd import capi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nbits", type=int, default=22)
    ap.add_argument("--queries", type=int, default=64)
    a = ap.parse_args()
    capi.init(0); capi.use_torch_stream()
    ne = a.nbits + 1
    steps = [ne]
    while steps[-1] > 6:
        steps.append(max(steps[-1] - 5, 5))
    B = GpuBackend(capi)
    flow.prove(B, 10, [11, 6], 4)                      # warm-up (tables, kernels)
    timers = {}
    t0 = time.perf_counter()
    out = flow.prove(B, a.nbits, steps, a.queries, seed=7, timers=timers)
    total = time.perf_counter() - t0
    print(json.dumps({"workload": f"toy genProof, N=2^{a.nbits}, FRI steps {steps}, {a.queries} queries, 1 GPU (interpreted step programs)",
                      "timers_s": {k: round(v, 5) for k, v in timers.items()}, "sum_timed_s": round(sum(timers.values()), 4),
                      "wall_s_incl_setup_and_transcript": round(total, 4), "root1": out["root1"]}))


if __name__ == "__main__":
    main()
