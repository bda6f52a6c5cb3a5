#!/usr/bin/env python3
"""One CHECKED run of Starks::genProof (starks.cpp:9-404) at a zkEVM-family geometry on the GPU: tools/bench_zkevm_genproof.py's
prove() with tests/zkevm_checks.py's oracle checks hooked in.  Prints one JSON line: the tool's report + "checks".
Run as a script (own process: at N = 2^23 the run needs ~293 of the 309 GB of HBM):
    python tests/zkevm_genproof_checked.py --circuit zkevm --log-n 23
TEST INFRASTRUCTURE (imports the oracle)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--circuit", default="zkevm")
    ap.add_argument("--log-n", type=int, default=0)
    ap.add_argument("--aliased", type=int, default=-1)
    ap.add_argument("--no-jit", action="store_true")
    ap.add_argument("--no-opt", action="store_true")
    ap.add_argument("--no-dots", action="store_true")
    ap.add_argument("--windows", type=int, default=16)
    b = ap.parse_args()
    from tools import bench_zkevm_genproof as T
    from tools import zkevm_geometry as G
    from tests.zkevm_checks import GenProofChecker
    from oracle import gl_oracle as orc
    from xlayer_prover_amd import capi
    capi.init(0)
    a = argparse.Namespace(log_n=b.log_n, circuit=b.circuit, batch=False, fri_steps="", queries=0, lookups=-1, zs=-1, no_jit=b.no_jit, no_dots=b.no_dots,
                           vm_only="", no_opt=b.no_opt, aliased=b.aliased)
    chk = GenProofChecker(orc, windows=b.windows)
    ln = b.log_n or G.CIRCUITS[b.circuit]["log_n"]
    out = T.prove(a, b.circuit, ln, hooks=chk)
    out["checks"] = chk.report
    print(json.dumps(out))


if __name__ == "__main__":
    main()
