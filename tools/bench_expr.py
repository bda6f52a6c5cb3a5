#!/usr/bin/env python3
"""Time the expression VM on the converted zkEVM programs (xlayer_prover_amd/chelpers/*.xlpexpr, built by
__graft_entry__.build() where the reference tree is mounted) or, without them, on a seeded random program.
Rows are processed in slabs of --rows (the packed synthetic slab replaces the 2^23-row one, tests/expr_real.py);
reports ms per slab and the extrapolation to the step's full domain (2^23 rows for *prev/step3, 2^24 for *2ns)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import expr_real as R
from xlayer_prover_amd import capi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-rows", type=int, default=17)
    ap.add_argument("--steps", default="step2prev,step3prev,step3,step42ns,step52ns")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--full", action="store_true", help="run every step on its full zkEVM domain (2^23 / 2^24 rows): measured, not extrapolated")
    ap.add_argument("--jit", action="store_true", help="attach the ahead-of-time plugin xlayer_prover_amd/chelpers/zkevm.<step>.so if present")
    a = ap.parse_args()
    capi.init(0); capi.use_torch_stream()
    out = {}
    for step in a.steps.split(","):
        log_rows = (24 if step.endswith("2ns") else 23) if a.full else a.log_rows
        n = 1 << log_rows
        path = os.path.join(R.REF_DIR, f"zkevm.{step}.xlpexpr")
        if not os.path.exists(path):
            print("missing", path); continue
        code, ntmp1, ntmp3 = R.read_program(path)
        code, info = R.shrink(code, n)
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        def fill(cnt, seed):
            t = torch.empty(cnt, dtype=torch.int64, device="cuda"); capi.dev_fill(t, seed, 0); return t
        dev = R.buffers(fill, 7, n, info)
        sp = capi.steps_params(n_const_pols=info["n_const"], n_const_pols_2ns=info["n_const2"], **dev)
        p = capi.ExprProgram(code, ntmp1, ntmp3)
        so = os.path.join(R.REF_DIR, f"zkevm.{step}.n{log_rows}.so")
        mode = "interpreter"
        if a.jit and os.path.exists(so):
            p.attach(so); mode = "compiled"
        p.run(sp, 0, n); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            p.run(sp, 0, n)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.reps * 1e3
        ninstr = sum(1 for k, s, *_ in R.operands(code) if s == 0)
        full = (1 << 24) if step.endswith("2ns") else (1 << 23)
        out[step] = {"mode": mode, "rows": n, "instructions": ninstr, "ntmp1": int(ntmp1), "ntmp3": int(ntmp3), "ms": round(ms, 3),
                     "ns_per_row_instr": round(ms * 1e6 / n / ninstr, 4), "full_domain_s": round(ms / 1e3 * full / n, 3)}
        p.close(); del dev, sp; torch.cuda.empty_cache()
        print(step, out[step], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
