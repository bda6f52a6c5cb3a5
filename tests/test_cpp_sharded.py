"""The ROW-SHARDED C++ Starks (host/starks.hpp genProofSharded; SURVEY.md 8e, BASELINE north_star) against the one-GPU C++ Starks: the same
synthetic witness, constants and plookup selections must give the SAME proof, bit for bit (zkin text), whatever the number of ranks.
One process per rank (host/test_host_shim `bundle ... --world G --rank r`), exchanges through the library's transport (csrc/comm.hip):
  * world 1, backend "self"  -- the window layout and every exchange code path with device copies;
  * world 1, backend "rccl"  -- the same through RCCL (librccl loaded, communicator of one rank, all-gathers and ring steps on it);
  * world 2 / 4, backend "shm" -- real ranks, host-staged because they share the test box's one GPU (RCCL wants one GPU per rank).
More than one GPU has never been available to this build; what is checked here is that the multi-rank orchestration is correct."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "xlayer_prover_amd", "host", "test_host_shim")
WSEED, CSEED = 0x5EED0401, 0x5EED0402


def _programs_built():
    from xlayer_prover_amd import programs as R
    return os.path.exists(os.path.join(R.REF_DIR, "zkevm.step52ns.xlpexpr"))


def test_window_bundle_is_consistent(tmp_path):
    """no GPU: the per-rank bundle of a sharded proof -- window offsets / rows in the starkinfo, the same views as the one-GPU starkinfo,
    next-row stores listed for the programs that have them"""
    from tools import make_stark_bundle as B
    from tools import zkevm_geometry as G
    if not _programs_built():
        pytest.skip("converted reference programs not built")
    ln, world = 12, 4
    out = B.make("zkevm", ln, False, out=str(tmp_path / "w"), world=world)
    j = json.load(open(os.path.join(out, "zkevm.starkinfo.json")))
    one = G.synth_starkinfo("zkevm", ln, False)
    L = G.layout_window(ln, world)
    assert j["xlp"]["world"] == world and j["xlp"]["haloN"] == 1 and j["xlp"]["haloExt"] == 2
    assert j["starkStruct"] == one["starkStruct"] and j["varPolMap"] == one["varPolMap"] and j["evMap"] == one["evMap"] and j["puCtx"] == one["puCtx"]
    n, ne = (1 << ln) // world, (1 << (ln + 1)) // world
    for name, off in j["mapOffsets"].items():
        assert off == L[name][0] and j["mapDeg"][name] == (ne + 2 if name.endswith("_2ns") else n + 1)
    assert j["mapTotalN"] == L["end"][0] < one["mapTotalN"]
    man = json.load(open(os.path.join(out, "zkevm.steps.json")))["steps"]
    assert man["step2prev"]["nextRowStores"] == [] and man["step42ns"]["nextRowStores"] == []
    st = man["step3prev"]["nextRowStores"]
    assert st and all(s[0] in ("cm3_n", "tmpExp_n") and s[3] == 1 for s in st)
    W = {"cm3_n": 408, "tmpExp_n": 389}
    assert all(s[1] + s[2] <= W[s[0]] for s in st)


def _run_ranks(bundle, world, backend, tmp_path, timeout=900):
    rdv = str(tmp_path / "rccl.id") if backend == "rccl" else f"xlp_test_{os.getpid()}_{world}"
    procs = []
    for r in range(world):
        cmd = [EXE, "bundle", bundle, "zkevm", str(WSEED), str(CSEED), "--world", str(world), "--rank", str(r)]
        if backend != "self":
            cmd += ["--backend", backend, "--rendezvous", rdv]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: {se[-3000:]}"
    return json.loads(outs[0][0]), [se for _, se in outs]


@pytest.mark.gpu
@pytest.mark.parametrize("world,backend", [(1, "self"), (1, "rccl"), (2, "shm"), (4, "shm")])
def test_cpp_sharded_starks_proof_equals_one_gpu_proof(tmp_path, world, backend):
    from tools import make_stark_bundle as B
    if not _programs_built():
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    ln = 12
    one = B.make("zkevm", ln, False, out=str(tmp_path / "one"))
    r = subprocess.run([EXE, "bundle", one, "zkevm", str(WSEED), str(CSEED)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    expect = json.loads(r.stdout)
    win = B.make("zkevm", ln, False, out=str(tmp_path / "win"), world=world)
    got, logs = _run_ranks(win, world, backend, tmp_path)
    assert "exchanges" in logs[0]
    for k in ("root1", "root2", "root3", "root4"):
        assert got[k] == expect[k], k
    assert got["evals"] == expect["evals"]
    assert got == expect                                              # every FRI tree, every opened row and sibling path, the final polynomial
