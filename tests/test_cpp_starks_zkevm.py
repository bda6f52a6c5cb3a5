"""The C++ entry points with the reference's shape -- Starks(config, StarkFiles, pAddress) + genProof(proof, publics, verkey, Steps*)
(host/starks.hpp; src/starkpil/starks.hpp:74,225) and Prover::genBatchProof (host/prover.hpp; src/prover/prover.cpp:392-669) --
driven at the zkEVM-family geometries from bundles (tools/make_stark_bundle.py: synthesized starkinfo + the converted reference
step programs).  Two independent orchestrations of the same kernels must print the same proof: the C++ Starks (witness in
page-locked host memory, band-pipelined step 1, StarkInfo-driven H1H2 / Z / evmap) and the Python flow of
tools/bench_zkevm_genproof.py (which tests/test_full_size.py checks against the oracle stage by stage); and the C++ proof is
verified on the oracle: transcript, all Merkle paths, all folds, final degree."""
import argparse
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "xlayer_prover_amd", "host")
EXE = os.path.join(HOST, "test_host_shim")
WSEED, CSEED = 0x5EED0401, 0x5EED0402


def _programs_built(circ):
    from xlayer_prover_amd import programs as R
    return os.path.exists(os.path.join(R.REF_DIR, f"{circ}.step52ns.xlpexpr"))


def test_bundle_starkinfo_is_consistent(tmp_path):
    """no GPU: the synthesized starkinfo of every circuit -- evaluation map entries name views that exist, the views lie inside their
    sections, cm_n lists the commit order (cm1 | h1, h2 per lookup | z per grand product | ...), offsets equal the slab layout"""
    from tools import zkevm_geometry as G
    for circ, ln, al in (("zkevm", 14, True), ("zkevm", 15, False), ("c12a", 20, False), ("recursive1", 17, False)):
        if not _programs_built(circ):
            pytest.skip("converted reference programs not built")
        j = G.synth_starkinfo(circ, ln, al)
        L = G.layout(ln, al, circ)
        W = j["mapSectionsN"]
        for v in j["varPolMap"]:
            assert v["sectionPos"] + v["dim"] <= W[v["section"]]
        for name, off in j["mapOffsets"].items():
            assert off == L[name][0]
        assert j["mapTotalN"] == L["end"][0]
        for e in j["evMap"]:
            if e["type"] == "cm":
                v = j["varPolMap"][j["cm_2ns"][e["id"]]]
                assert v["section"].endswith("_2ns")
            elif e["type"] == "q":
                assert j["varPolMap"][j["qs"][e["id"]]]["section"] == "cm4_2ns"
            else:
                assert e["id"] < j["nConstants"]
        n1, npu = j["nCm1"], len(j["puCtx"])
        for i, c in enumerate(j["puCtx"]):
            assert j["cm_n"][n1 + 2 * i] == c["h1Id"] and j["varPolMap"][c["h1Id"]] == {"section": "cm2_n", "dim": 3, "sectionPos": 6 * i}
        zs = [c["zId"] for c in j["puCtx"]] + [c["zId"] for c in j["peCtx"]]
        for k, z in enumerate(zs):
            assert j["cm_n"][n1 + 2 * npu + k] == z and j["varPolMap"][z] == {"section": "cm3_n", "dim": 3, "sectionPos": 3 * k}
        assert (j["xlp"]["const2nsOffset"] is not None) == al
        if circ == "recursive1":
            assert len(j["evMap"]) == 118 and j["nPublics"] == 48 and j["nConstants"] == 52      # the golden recursive1 proofs' shape


def _flow(circ, ln, aliased):
    import torch
    from tools import bench_zkevm_genproof as T
    from xlayer_prover_amd import capi
    capi.init(0)
    a = argparse.Namespace(log_n=ln, circuit=circ, batch=False, fri_steps="", queries=0, lookups=-1, zs=-1, no_jit=False, no_dots=False, vm_only="", no_opt=False,
                           aliased=int(aliased))
    out = T.prove(a, circ, ln)
    torch.cuda.empty_cache()
    return out


def _cpp_bundle(bundle, circ, timeout=360):
    r = subprocess.run([EXE, "bundle", bundle, circ, str(WSEED), str(CSEED)], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout), r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("circ,ln,aliased", [("zkevm", 14, True), ("recursive1", 17, False), ("c12a", 20, False)])     # (the unaliased zkEVM slab: tests/test_cpp_sharded.py's one-GPU reference, tests/test_full_size.py)
def test_cpp_starks_equals_python_flow_and_verifies(orc, tmp_path, circ, ln, aliased):
    import torch
    from tools import make_stark_bundle as B
    from tools import zkevm_geometry as G
    from tests.zkin_verify import verify_zkin, ints
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if not _programs_built(circ):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    bundle = B.make(circ, ln, aliased, out=str(tmp_path / "bundle"))
    Z, log = _cpp_bundle(bundle, circ)
    assert "dot_columns" in log                                   # the FRI polynomial ran as table program + dot_columns + row program
    exp = _flow(circ, ln, aliased)
    # ---- same proof as the Python orchestration
    assert [ints(Z[k]) for k in ("root1", "root2", "root3", "root4")] == exp["roots"]
    fr = exp["fri_roots_and_head_of_final_pol"]
    si = G.synth_starkinfo(circ, ln, aliased)
    steps = [s["nBits"] for s in si["starkStruct"]["steps"]]
    for k in range(1, len(steps)):
        assert ints(Z[f"s{k}_root"]) == fr[k - 1]
    assert [x for e in Z["finalPol"] for x in ints(e)][:6] == fr[-1]
    # ---- and a valid one
    W = si["mapSectionsN"]
    widths = {"1": W["cm1_2ns"], "2": W["cm2_2ns"], "3": W["cm3_2ns"], "4": W["cm4_2ns"], "C": si["nConstants"]}
    rep = verify_zkin(orc, Z, ln, ln + G.CIRCUITS[circ]["ext"], steps, si["starkStruct"]["nQueries"], widths)
    assert rep["paths"] > 0 and rep["folds"] == si["starkStruct"]["nQueries"] * (len(steps) - 1)
    assert len(Z["evals"]) == len(si["evMap"])


def _batch_config(tmp_path, bundles):
    cfg = {"runFileGenBatchProof": True, "saveOutputToFile": True, "saveProofToFile": True, "outputPath": str(tmp_path), "xlpSyntheticLookups": True}
    for name, b in bundles.items():
        cfg[name + "StarkInfo"] = os.path.join(b, f"{name}.starkinfo.json")
        cfg[name + "ConstPols"] = f"synthetic:{CSEED + len(name)}"
        cfg[name + "ConstantsTree"] = ""
        cfg[name + "Verkey"] = ""
    f = tmp_path / "config.json"
    f.write_text(json.dumps(cfg))
    return str(f)


@pytest.mark.gpu
def test_genbatchproof_chains_three_starks(orc, tmp_path):
    """Prover::genBatchProof (host/prover.hpp) on small bundles: zkEVM programs at 2^13 rows, c12a and recursive1 at their real
    geometry.  The recursive1 proof it returns verifies; its publics are the zkEVM witness's first / last rows at the fork-9 Main
    register columns; both upload modes give the same proof; another executor seed changes every proof down the chain."""
    import torch
    from tools import make_stark_bundle as B
    from tools import zkevm_geometry as G
    from tests.zkin_verify import verify_zkin, ints
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    for c in ("zkevm", "c12a", "recursive1"):
        if not _programs_built(c):
            pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    bundles = {"zkevm": B.make("zkevm", 13, True, out=str(tmp_path / "zkevm")), "c12a": B.make("c12a", 20, False, out=str(tmp_path / "c12a")),
               "recursive1": B.make("recursive1", 17, False, out=str(tmp_path / "recursive1"))}
    cfg = _batch_config(tmp_path, bundles)
    exe = os.path.join(HOST, "zkprover_batch")

    def run(*extra):
        r = subprocess.run([exe, "-c", cfg, "--proofs", "1"] + list(extra), capture_output=True, text=True, timeout=360)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads(r.stdout.strip().splitlines()[-1]), json.load(open(tmp_path / "batch_proof.output.json")), json.load(open(tmp_path / "batch_proof.public.json"))
    line, Z, pub = run()
    assert line["three_starks_s"] > 0 and line["zkevm_witness_upload_ms"] > 0
    for k in ("STARK_PROOF_BATCH_PROOF", "STARK_C12_A_PROOF_BATCH_PROOF", "STARK_RECURSIVE_1_PROOF_BATCH_PROOF", "zkevm.STARK_STEP_1_LDE_AND_MERKLETREE", "recursive1.STARK_STEP_FRI"):
        assert k in line["timers_s"]
    # publics: rows 0 and N-1 of the synthetic zkEVM witness at the Main register columns (prover.cpp:494-559)
    N, Wz = 1 << 13, 751
    at = lambda row, col: int(orc.fill(1, WSEED, row * Wz + col)[0])
    assert ints(pub[:8]) == [at(0, 581 - k) for k in range(8)] and ints(pub[19:27]) == [at(N - 1, 613 - k) for k in range(8)]
    assert int(pub[16]) == at(0, 615) and int(pub[43]) == at(N - 1, 616) and ints(pub[44:48]) == [0, 0, 0, 0]
    assert Z["publics"] == pub
    si = G.synth_starkinfo("recursive1", 17, False)
    steps = [s["nBits"] for s in si["starkStruct"]["steps"]]
    W = si["mapSectionsN"]
    # the verification key is not in the zkin; the recursive1 Starks printed none -> recompute it the way the transcript needs it: replay with
    # each candidate is not possible, so take it from a bundle run of the same constants
    ZZ, _ = _cpp_bundle_verkey(bundles["recursive1"], cfg)
    Z["verkey"] = ZZ
    rep = verify_zkin(orc, Z, 17, 20, steps, 43, {"1": W["cm1_2ns"], "2": 0, "3": W["cm3_2ns"], "4": W["cm4_2ns"], "C": si["nConstants"]})
    assert rep["folds"] == 43 * 4
    line2, Z2, _ = run("--sequential-upload")
    assert line2["batch_proof_fnv"] == line["batch_proof_fnv"] and "zkevm.STARK_STEP_1_WITNESS_UPLOAD" in line2["timers_s"]
    line3, Z3, pub3 = run("--seed", "77")
    assert line3["batch_proof_fnv"] != line["batch_proof_fnv"] and Z3["root1"] != Z2["root1"] and pub3 != pub
    # three proofs in one process: in the aliased layout the extended constants are recomputed in step 4 of a proof; from the second proof
    # on a resident copy is taken when the memory is there (here it is; at 2^23 rows on 288 GB it is not: 10.5 GB free at that point) and
    # the third proof no longer extends them -- the proofs stay the same
    r = subprocess.run([exe, "-c", cfg, "--proofs", "3"], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.strip().splitlines() if l.startswith("{") and "timers_s" in l]
    assert len(lines) == 3 and all(l["batch_proof_fnv"] == line["batch_proof_fnv"] for l in lines)
    assert "zkevm.STARK_STEP_4_CONST_LDE" in lines[0]["timers_s"] and "zkevm.STARK_STEP_4_CONST_LDE" in lines[1]["timers_s"]
    assert "zkevm.STARK_STEP_4_CONST_LDE" not in lines[2]["timers_s"]


def _cpp_bundle_verkey(bundle, cfg_file):
    """constRoot of the recursive1 constants named by the batch config (synthetic:<seed>): a bundle run with that seed prints it"""
    cfg = json.load(open(cfg_file))
    seed = int(cfg["recursive1ConstPols"].split(":")[1])
    r = subprocess.run([EXE, "bundle", bundle, "recursive1", "1", str(seed)], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout)["verkey"], None


@pytest.mark.gpu
def test_e2e_chain_batch_aggregated_final(orc, tmp_path):
    """BASELINE configs[4] in C++ (host/prover.hpp through `zkprover_batch --e2e`; the SNARK wrap on a synthetic final circuit): two consecutive batch proofs
    (zkEVM programs at 2^13 rows; c12a, recursive1 at their real geometry) -> Prover::genAggregatedProof (joinzkin + the recursive2
    STARK) -> Prover::genFinalProof up to the recursivef STARK (real geometry 2^17 x 16, Poseidon-BN128 trees and transcript).
    The aggregated proof verifies on the Goldilocks oracle and carries the joined publics; the recursivef proof verifies on the BN128
    oracle (TranscriptBN128 replay, MerkleTreeBN128 openings, FRI folds, final degree)."""
    import torch
    from oracle import bn128_oracle as bo
    from tools import make_stark_bundle as B
    from tools import zkevm_geometry as G
    from tests.zkin_verify import verify_zkin, verify_zkin_bn128, ints
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    const = os.path.join(ROOT, "xlayer_prover_amd", "chelpers", "poseidon_bn128.bin")
    for c in ("zkevm", "c12a", "recursive1", "recursive2", "recursivef"):
        if not _programs_built(c):
            pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    if not os.path.exists(const):
        pytest.skip("poseidon_bn128.bin not generated")
    geo = {"zkevm": (13, True), "c12a": (20, False), "recursive1": (17, False), "recursive2": (17, False), "recursivef": (17, False)}
    bundles = {c: B.make(c, ln, al, out=str(tmp_path / c)) for c, (ln, al) in geo.items()}
    cfg = _batch_config(tmp_path, bundles)
    # the SNARK wrap: an fflonk key for a synthetic final circuit (known tau, oracle/fflonk_oracle.py) as config.finalStarkZkey, its witness
    # through --final-wtns in place of the final circuit's witness calculator
    from oracle import fflonk_oracle as FF
    zk, wit, tau = FF.setup(6, n_public=1, n_additions=2, seed=31)
    FF.write_zkey(str(tmp_path / "final.fflonk.zkey"), zk)
    FF.write_wtns(str(tmp_path / "final.wtns"), wit)
    cj = json.load(open(cfg))
    cj["finalStarkZkey"] = str(tmp_path / "final.fflonk.zkey")
    json.dump(cj, open(cfg, "w"))
    exe = os.path.join(HOST, "zkprover_batch")
    r = subprocess.run([exe, "-c", cfg, "--e2e", "--final-wtns", str(tmp_path / "final.wtns")], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.strip().splitlines()]
    assert [l["stage"] for l in lines] == ["genBatchProof#1", "genBatchProof#2", "genAggregatedProof", "genFinalProof(recursivef)", "finalProof(fflonk)"]
    assert "RAPID_SNARK" in lines[3]["timers_s"] and "fflonk.FFLONK_ROUND_5" in lines[3]["timers_s"]
    assert FF.verify(zk, tau, lines[4]["proof"], lines[4]["publics"]) and lines[4]["publics"] == [str(wit[1])]
    assert lines[0]["output_fnv"] != lines[1]["output_fnv"]
    assert "STARK_RECURSIVE_2_PROOF_BATCH_PROOF" in lines[2]["timers_s"] and "STARK_RECURSIVE_F_PROOF_BATCH_PROOF" in lines[3]["timers_s"]
    # ---- the aggregated (recursive2) proof
    Z = json.load(open(tmp_path / "aggregated_proof.output.json"))
    Z["verkey"] = lines[2]["verkey"]
    si = G.synth_starkinfo("recursive2", 17, False)
    steps = [s["nBits"] for s in si["starkStruct"]["steps"]]
    W = si["mapSectionsN"]
    rep = verify_zkin(orc, Z, 17, 20, steps, si["starkStruct"]["nQueries"], {"1": W["cm1_2ns"], "2": 0, "3": W["cm3_2ns"], "4": W["cm4_2ns"], "C": si["nConstants"]})
    assert rep["folds"] == si["starkStruct"]["nQueries"] * (len(steps) - 1)
    assert len(Z["publics"]) == 48 and ints(Z["publics"][44:48]) == ints(lines[2]["verkey"])       # joined publics + recursive2's own constRoot
    # ---- the recursivef proof
    ZF = json.load(open(tmp_path / "recursivef.zkin.json"))
    sf = G.synth_starkinfo("recursivef", 17, False)
    assert sf["starkStruct"]["verificationHashType"] == "BN128"
    stepsf = [s["nBits"] for s in sf["starkStruct"]["steps"]]
    Wf = sf["mapSectionsN"]
    K = bo.Constants(const)
    repf = verify_zkin_bn128(orc, bo, K, ZF, 17, 21, stepsf, sf["starkStruct"]["nQueries"],
                             {"1": Wf["cm1_2ns"], "2": 0, "3": Wf["cm3_2ns"], "4": Wf["cm4_2ns"], "C": sf["nConstants"]})
    assert repf["verkey"] == int(lines[3]["verkey"])                                                 # rebuilt from the first opening == the prover's constant root
    assert repf["folds"] == sf["starkStruct"]["nQueries"] * (len(stepsf) - 1) and repf["paths"] >= 32 * 8
    assert ZF["publics"] == Z["publics"][:len(ZF["publics"])] and ZF["aggregatorAddr"] == "0"
