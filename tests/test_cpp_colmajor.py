"""COLUMN-MAJOR device sections in the C++ product path (host/starks.hpp with a "colMajor" starkinfo; include/xlp_gl.h): the proof must be the
row-major run's proof, bit for bit -- the same witness, constants and plookup selections, every stage on the other layout (witness bands
transposed on arrival, step programs addressing columns and cosets, extendPol / leaf sponge / evmap / FRI-polynomial sums on columns, opened rows
gathered).  One process per run (host/test_host_shim `bundle ...`, host/zkprover_batch)."""
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


def _proof(bundle, circ="zkevm"):
    r = subprocess.run([EXE, "bundle", bundle, circ, str(WSEED), str(CSEED)], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout), r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ln,aliased,use_opt,use_dots", [(12, False, False, False), (13, True, True, True)])
def test_colmajor_zkevm_proof_equals_rowmajor_proof(tmp_path, ln, aliased, use_opt, use_dots):
    """THE layout-equivalence test (round 5: the row-major device orchestration of host/starks.hpp is kept for this comparison and for StarkRecursiveF only;
    no row-major bundle is built or benchmarked any more).  The zkEVM STARK at 2^12 / 2^13 rows: programs as converted / in dot-product form with the FRI
    polynomial on xlp_dev_dot_columns_cm; the unaliased and the one-GPU aliased slab (extended constants inside the slab, re-extended per proof)"""
    from tools import make_stark_bundle as B
    if not _programs_built():
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    row = B.make("zkevm", ln, aliased, out=str(tmp_path / "row"), use_opt=use_opt, use_dots=use_dots, colmajor=False)
    col = B.make("zkevm", ln, aliased, out=str(tmp_path / "col"), use_opt=use_opt, use_dots=use_dots, colmajor=True)
    assert json.load(open(os.path.join(col, "zkevm.starkinfo.json")))["xlp"]["colMajor"] is True
    expect, _ = _proof(row)
    got, _ = _proof(col)
    for k in ("root1", "root2", "root3", "root4"):
        assert got[k] == expect[k], k
    assert got["evals"] == expect["evals"]
    assert got == expect


@pytest.mark.gpu
def test_genbatchproof_pipelined_upload_equals_sequential_upload(tmp_path):
    """Prover::genBatchProof (host/zkprover_batch) on column-major bundles: the witness upload pipelined with step 1 == uploaded first (FNV of the batch
    proof's zkin text), the second proof of a process == the first (resident constants, reused scratch), and the LDE / Merkle stage timers are there.
    (The batch proof against ROW-MAJOR sections was compared here through round 4; the layouts are now compared once, above.)"""
    from tools import make_stark_bundle as B
    if not _programs_built():
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    exe = os.path.join(ROOT, "xlayer_prover_amd", "host", "zkprover_batch")
    bundles = {"zkevm": B.make("zkevm", 14, True, out=str(tmp_path / "zk")), "c12a": B.make("c12a", 20, False, out=str(tmp_path / "c12a")),
               "recursive1": B.make("recursive1", 17, False, out=str(tmp_path / "r1"))}
    cfg = {"runFileGenBatchProof": True, "saveOutputToFile": False, "saveProofToFile": False, "outputPath": str(tmp_path), "xlpSyntheticLookups": True}
    for n, b in bundles.items():
        cfg[n + "StarkInfo"] = os.path.join(b, f"{n}.starkinfo.json")
        cfg[n + "ConstPols"] = f"synthetic:{CSEED + len(n)}"
        cfg[n + "ConstantsTree"] = ""
        cfg[n + "Verkey"] = ""
    f = tmp_path / "cfg.json"
    f.write_text(json.dumps(cfg))
    outs = []
    for extra in ([], ["--sequential-upload"]):
        r = subprocess.run([exe, "-c", str(f), "--proofs", "2"] + extra, capture_output=True, text=True, timeout=360)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [json.loads(l) for l in r.stdout.strip().splitlines()]
        assert lines[0]["batch_proof_fnv"] == lines[1]["batch_proof_fnv"]
        outs.append(lines[-1])
    assert outs[0]["batch_proof_fnv"] == outs[1]["batch_proof_fnv"] and outs[0]["batch_proof_bytes"] == outs[1]["batch_proof_bytes"]
    t = outs[0]["timers_s"]
    assert "zkevm.STARK_STEP_1_LDE" in t and "zkevm.STARK_STEP_1_MERKLETREE" in t


@pytest.mark.gpu
def test_commit_file_round_trip_through_the_prover(tmp_path):
    """config.zkevmCmPolsAfterExecutor (prover.cpp:452-457): the executor's output is written as the reference's zkevm.commit (row-major,
    pilDegree x nCommitted words); fed back with --witness-file in place of the executor it gives the same batch proof"""
    from tools import make_stark_bundle as B
    if not _programs_built():
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    exe = os.path.join(ROOT, "xlayer_prover_amd", "host", "zkprover_batch")
    bundles = {"zkevm": B.make("zkevm", 12, True, out=str(tmp_path / "zk"), colmajor=True), "c12a": B.make("c12a", 20, False, out=str(tmp_path / "c12a"), colmajor=True),
               "recursive1": B.make("recursive1", 17, False, out=str(tmp_path / "r1"), colmajor=True)}
    commit = str(tmp_path / "zkevm.commit")

    def config(name, dump):
        cfg = {"runFileGenBatchProof": True, "saveOutputToFile": False, "saveProofToFile": False, "outputPath": str(tmp_path), "xlpSyntheticLookups": True}
        if dump:
            cfg["zkevmCmPolsAfterExecutor"] = commit
        for n, b in bundles.items():
            cfg[n + "StarkInfo"] = os.path.join(b, f"{n}.starkinfo.json")
            cfg[n + "ConstPols"] = f"synthetic:{CSEED + len(n)}"
            cfg[n + "ConstantsTree"] = ""
            cfg[n + "Verkey"] = ""
        f = tmp_path / name
        f.write_text(json.dumps(cfg))
        return str(f)
    r = subprocess.run([exe, "-c", config("dump.json", True), "--proofs", "1"], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr[-3000:]
    first = json.loads(r.stdout.strip().splitlines()[-1])
    w = json.load(open(os.path.join(bundles["zkevm"], "zkevm.starkinfo.json")))["mapSectionsN"]["cm1_n"]
    assert os.path.getsize(commit) == (1 << 12) * w * 8
    r = subprocess.run([exe, "-c", config("read.json", False), "--proofs", "1", "--seed", "12345", "--witness-file", commit], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr[-3000:]
    again = json.loads(r.stdout.strip().splitlines()[-1])
    assert again["batch_proof_fnv"] == first["batch_proof_fnv"] and again["batch_proof_bytes"] == first["batch_proof_bytes"]


@pytest.mark.gpu
def test_merkelize_memory_is_the_merkle_root_of_the_slab_words(tmp_path, orc):
    """Starks::merkelizeMemory (starks.cpp:671-700; host/starks.hpp): the reference's debug hash of the polynomial memory -- the slab's words as a matrix of
    2^k rows (k as the reference's loop finds it), merkletree, the root -- over the device slab after a proof == the oracle's merkletree of the same words"""
    import numpy as np
    from tools import make_stark_bundle as B
    if not _programs_built():
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    col = B.make("zkevm", 10, False, out=str(tmp_path / "col"), colmajor=True)
    dump = str(tmp_path / "slab.bin")
    r = subprocess.run([EXE, "bundle", col, "zkevm", str(WSEED), str(CSEED), "--merkelize", dump], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stderr.splitlines() if l.startswith("rootDBG ")][-1].split()
    root, words = [int(x) for x in line[1:5]], int(line[6])
    slab = np.fromfile(dump, dtype=np.uint64)
    assert slab.size == words and words % 2 == 0
    nrows = 2
    while words % (nrows * 2) == 0:
        nrows *= 2
    ncols = words // nrows
    assert ncols > 4                                          # (a real sponge per row, not the copy rule of narrow trees)
    nodes = orc.merkletree(slab.reshape(nrows, ncols), ncols, nrows)
    assert [int(x) for x in nodes[-4:]] == root
