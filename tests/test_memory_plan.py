"""The memory plan of a row-sharded batch proof at the real size: host/zkprover_batch walks Prover::genBatchProof ->
Starks::genProofSharded for ONE rank of G at 2^23 rows in PLAN MODE (XLP_PLAN=1, include/xlp_gl.h: no GPU is touched, none needs to exist; device
allocations are accounted by the code that makes them -- the window slab, the replicated constants and their tree, the trees of the sections, the two
staging sets of the sharded commit, the x / Zi tables, the library's scratch slots and twiddle tables) and prints what that rank would hold at its peak.
No box of the build pool has more than one GPU, so this is the evidence that the N > 1 path FITS: 288 GiB per MI355X, with the margin the one-GPU path
keeps.  CPU only."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "xlayer_prover_amd", "host", "zkprover_batch")
BUNDLES = os.path.join(ROOT, "xlayer_prover_amd", "chelpers", "bundles")
GIB = 1 << 30
HBM = 288 * GIB
MARGIN = 8 * GIB          # what the one-GPU path leaves free at its peak on a real MI355X (its slab alone is 239 GB: profiles/r4/README.md)


def _plan(tmp_path, world, rank, proofs=2):
    name = f"zkevm.geo23w{world}c" if world else "zkevm.geo23ac"
    si = os.path.join(BUNDLES, name, "zkevm.starkinfo.json")
    if not (os.path.exists(EXE) and os.path.exists(si) and os.path.exists(os.path.join(BUNDLES, "c12a.geo20c", "c12a.starkinfo.json"))):
        pytest.skip("bundles / host binary not built (needs the reference tree at build time)")
    cfg = json.load(open(os.path.join(ROOT, "tools", "config_runFile_BatchProof_synthetic_colmajor.json")))
    cfg["zkevmStarkInfo"] = si
    for k in cfg:
        if k.endswith("StarkInfo") and not os.path.isabs(cfg[k]):
            cfg[k] = os.path.join(ROOT, cfg[k])
    f = tmp_path / f"cfg_{world}_{rank}.json"
    f.write_text(json.dumps(cfg))
    env = dict(os.environ, XLP_PLAN="1")
    args = [EXE, "-c", str(f), "--proofs", str(proofs)]
    if world:
        args += ["--world", str(world), "--rank", str(rank), "--backend", "shm", "--rendezvous", f"xlp_plan_{os.getpid()}_{world}_{rank}"]
    r = subprocess.run(args, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    plans = [json.loads(l)["memory_plan"] for l in r.stdout.splitlines() if l.startswith('{"memory_plan"')]
    assert len(plans) == proofs
    return plans


@pytest.mark.parametrize("world", [2, 4, 8])
def test_a_rank_of_the_sharded_batch_proof_at_2p23_fits_one_mi355x(tmp_path, world):
    """first and last rank of G (the last rank's halo wraps; rank 0 also proves c12a and recursive1 afterwards): the peak over two consecutive batch proofs
    (the second one meets the grow-only scratch of the first) stays below 288 GiB minus the margin, the window slab is what dominates, and the peak
    shrinks as the ranks grow"""
    peaks = {}
    for rank in (0, world - 1):
        plans = _plan(tmp_path, world, rank)
        peak = max(p["peak_bytes"] for p in plans)
        assert peak <= HBM - MARGIN, (world, rank, peak / GIB)
        assert plans[-1]["largest_at_peak"][0] >= 0.25 * peak         # the window slab: one allocation, the largest (a quarter of the peak at G = 8, half at G = 2)
        assert plans[0]["world"] == world and plans[0]["rank"] == rank
        peaks[rank] = peak
    # measured plan (round 5): 256 / 173 / 131 GiB at G = 2 / 4 / 8 (one GPU: 266 GiB) -- the window slab halves with G, the replicated constants
    # (44 GiB), the staging sets (24-27 GiB) and the scratch slots (14 GiB) do not: a regression of more than a few GiB fails here
    assert max(peaks.values()) < {2: 262, 4: 180, 8: 136}[world] * GIB, {k: v / GIB for k, v in peaks.items()}


def test_plan_of_the_one_gpu_batch_proof_matches_what_the_real_runs_need(tmp_path):
    """calibration of the account against reality: the one-GPU batch proof (aliased 239 GB slab) is known to run on a 288 GiB MI355X with a few GB to
    spare (every bench of rounds 2-5); its plan must say the same -- between 270 and 288 GiB"""
    plans = _plan(tmp_path, 0, 0)
    peak = max(p["peak_bytes"] for p in plans)
    assert 255 * GIB < peak <= HBM, peak / GIB
