#!/usr/bin/env python3
"""Stage timers of the C++ Starks at the zkEVM geometry, one GPU: the one-GPU genProof next to the ROW-SHARDED genProofSharded
(host/starks.hpp) in a world of one (backend self and RCCL) and as 2 / 4 ranks SHARING this GPU (backend shm: every exchange staged
through the host).  What it shows: (1) the same proof comes out (zkin text hash), (2) what the sharded orchestration itself costs when
nothing is gained from it (a world of one: the window copies, exchanges that are device copies, gathered 3-column polynomials), which is
the overhead term of the scaling model in DESIGN.md 6.  The multi-rank rates mean nothing (one GPU, host-staged); no multi-GPU box exists
in this pool.  Step programs run interpreted unless ahead-of-time kernels of this geometry were built
(`python tools/build_expr_plugins.py --geometry L [--world G] [--colmajor] --opt-only` in xlayer_prover_amd/chelpers).
--layouts rowmajor,colmajor: every configuration with row-major sections / windows and with column-major ones (round 4).
    python tools/bench_sharded_cpp.py [--log-n 20] [--worlds 1,2,4] [--layouts rowmajor,colmajor]"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "xlayer_prover_amd", "host", "test_host_shim")
WSEED, CSEED = "0x5EED0401", "0x5EED0402"


def timers(stderr):
    return {m.group(1): float(m.group(2)) for m in re.finditer(r"<-- (\S+) done: ([0-9.]+) s", stderr)}


def stage_sum(t):
    """the stages as the wall sees them: STARK_STEP_n_LDE / _MERKLETREE (n = 1..3) are kernel-event sums INSIDE STARK_STEP_n_LDE_AND_MERKLETREE"""
    inner = {f"STARK_STEP_{n}_{h}" for n in (1, 2, 3) for h in ("LDE", "MERKLETREE")}
    return sum(v for k, v in t.items() if k not in inner)


def run(bundle, world=0, backend="self", tag=""):
    procs = []
    rdv = os.path.join(tempfile.gettempdir(), f"xlp_bsc_{os.getpid()}_{tag}") if backend == "rccl" else f"xlp_bsc_{os.getpid()}_{tag}"
    for r in range(max(world, 1)):
        cmd = [EXE, "bundle", bundle, "zkevm", WSEED, CSEED, "timers"]
        if world:
            cmd += ["--world", str(world), "--rank", str(r)]
            if backend != "self":
                cmd += ["--backend", backend, "--rendezvous", rdv]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, XLP_COMM_BARRIER_TIMEOUT_S="200")))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300))
    except subprocess.TimeoutExpired:
        for p in procs:
            if p.poll() is None:
                p.kill()
        tails = [p.communicate()[1][-300:] for p in procs]
        print(f"[bench_sharded_cpp] {os.path.basename(bundle)} world {world} {backend}: no result within 300 s", file=sys.stderr, flush=True)
        return {"error": "no result within 300 s", "stderr_tails": tails}
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        if p.returncode != 0:
            return {"error": f"rank {r}: exit {p.returncode}: {se[-400:]}"}
    t = timers(outs[0][1])
    print(f"[bench_sharded_cpp] {os.path.basename(bundle)} world {world} {backend}: {stage_sum(t):.3f} s of timed stages", file=sys.stderr, flush=True)
    modes = f"{outs[0][1].count('compiled')} compiled"
    ex = re.search(r"(\d+) exchanges, ([0-9.]+) GB sent to peers in ([0-9.]+) s", outs[0][1])
    return {"proof_sha256_16": hashlib.sha256(outs[0][0].strip().splitlines()[-1].encode()).hexdigest()[:16], "sum_timed_s": round(stage_sum(t), 4),
            "timers_s_rank0": {k: round(v, 4) for k, v in t.items()}, "programs": modes,
            "exchanges_rank0": {"calls": int(ex.group(1)), "GB_sent_to_peers": float(ex.group(2)), "seconds": float(ex.group(3))} if ex else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--worlds", default="1,2,4")
    ap.add_argument("--layouts", default="rowmajor")
    a = ap.parse_args()
    from tools import make_stark_bundle as B
    tmp = tempfile.mkdtemp(prefix="xlp_bsc_")
    out = {"workload": f"zkEVM-geometry genProof (C++ Starks), N = 2^{a.log_n}, ONE GPU", "runs": {}}
    one = None
    for layout in a.layouts.split(","):
        cm = layout == "colmajor"
        sfx = "_colmajor" if cm else ""
        r1 = run(B.make("zkevm", a.log_n, False, out=os.path.join(tmp, "one" + sfx), colmajor=cm))
        one = one or r1
        out["runs"]["one_gpu" + sfx] = r1
        for w in [int(x) for x in a.worlds.split(",")]:
            b = B.make("zkevm", a.log_n, False, out=os.path.join(tmp, f"w{w}{sfx}"), world=w, colmajor=cm)
            if w == 1:
                out["runs"]["sharded_world1_self" + sfx] = run(b, 1, "self", "w1s")
                out["runs"]["sharded_world1_rccl" + sfx] = run(b, 1, "rccl", "w1r")
            else:
                out["runs"][f"sharded_world{w}_shm_ranks_sharing_the_gpu" + sfx] = run(b, w, "shm", f"w{w}")
    ref = one.get("proof_sha256_16")
    out["same_proof_everywhere"] = all(r.get("proof_sha256_16") == ref for r in out["runs"].values())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
