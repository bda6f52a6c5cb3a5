#!/usr/bin/env python3
"""Writes the mini PIL's files (tests/mini_genproof_flow.py) for N = 2^nbits and runs the C++ Starks::genProof mirror
(xlayer_prover_amd/host/test_host_shim genproof ... timers); prints its stage timers."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import mini_genproof_flow as flow
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ne = nbits + 1
steps = [ne]
while steps[-1] > 6:
    steps.append(max(steps[-1] - 5, 5))
d = tempfile.mkdtemp()
flow.toy_stark_files(d, "toy", nbits, steps, 32, seed=1)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xlayer_prover_amd", "host", "test_host_shim")
t0 = time.perf_counter()
r = subprocess.run([exe, "genproof", d, "toy", "timers"], capture_output=True, text=True)
print(r.stderr.strip())
print(f"process wall (incl. file IO, setup, constant tree): {time.perf_counter() - t0:.3f} s; rc={r.returncode}; proof bytes={len(r.stdout)}")
