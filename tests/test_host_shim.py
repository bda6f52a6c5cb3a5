"""The C++ host-side mirror of the reference interface (xlayer_prover_amd/host):
a mini Starks::genProof (commit stages + FRIProve::prove) driven through the
reference-shaped classes over the C ABI, compared field by field with the same
proof rebuilt by the CPU oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "xlayer_prover_amd", "host", "test_host_shim")


def test_host_shim_is_built_and_fails_loudly_without_gpu():
    import torch
    assert os.path.exists(EXE), "run __graft_entry__.build()"
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([EXE, "4", "5", "2", "5", "3"], capture_output=True, text=True)
    assert r.returncode != 0                       # exitProcess() semantics, no CPU fallback
    assert "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(6, 7, 4, [7, 4]), (10, 11, 8, [11, 8, 5]), (9, 12, 11, [12, 8, 4]), (8, 9, 5, [9, 9, 6, 3])])
def test_mini_genproof_matches_oracle(orc, cfg):
    from tests.mini_stark_oracle import mini_stark_oracle
    nBits, nBitsExt, nQ, steps = cfg
    r = subprocess.run([EXE, str(nBits), str(nBitsExt), str(nQ)] + [str(s) for s in steps], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    exp = mini_stark_oracle(orc, nBits, nBitsExt, nQ, steps)
    assert set(got.keys()) == set(exp.keys())
    for k in exp:
        assert got[k] == exp[k], k
    # zkin geometry matches the reference's golden files (proof2zkinStark.cpp:8-82): absent cm2 keys
    assert "s0_vals2" not in got and "s0_siblings2" not in got
    assert len(got["s0_siblings1"][0]) == nBitsExt and len(got["finalPol"]) == 1 << steps[-1]


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed", [(5, 1), (64, 2), (1000, 3)])
def test_polinomial_mirror_matches_oracle(orc, n, seed):
    """Polinomial::batchInverse / batchInverseParallel / calculateZ / calculateH1H2_opt1 of the C++ mirror (strided
    views of one slab, polinomial.hpp:249-720) against the oracle"""
    import numpy as np
    r = subprocess.run([EXE, "polinomial", str(n), str(seed)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {k: np.array([int(v) for v in vals], np.uint64) for k, vals in json.loads(r.stdout).items()}
    slab = got["slab"].reshape(n, 11)
    a, b = np.ascontiguousarray(slab[:, 0:3]), np.ascontiguousarray(slab[:, 3:6])
    assert (got["inv3"].reshape(n, 3) == orc.batch_inverse(a.reshape(-1), 3)).all()
    assert (got["z"].reshape(n, 3) == orc.calculate_z(a, b)).all()
    assert (got["inv1"] == orc.batch_inverse(np.ascontiguousarray(slab[:, 9]), 1)).all()
    h1, h2 = orc.calculate_h1h2(got["f"], got["t"], 1)
    assert (got["h1"] == h1).all() and (got["h2"] == h2).all()


def _zkevm_like_starkinfo():
    """a starkinfo.json with the fork-9 zkEVM geometry: section widths 751 / 168 / 408 / 6 / 389, N = 2^23, blow-up 2.
    The offsets computed here the way pil-stark lays the slab out (sections in eSection order of use: cm*_n, tmpExp_n,
    cm*_2ns, q_2ns, f_2ns) must reproduce the offsets baked into the reference's generated step tables."""
    N, NE = 1 << 23, 1 << 24
    widths = {"cm1_n": 751, "cm2_n": 168, "cm3_n": 408, "cm4_n": 6, "tmpExp_n": 389, "cm1_2ns": 751, "cm2_2ns": 168, "cm3_2ns": 408, "cm4_2ns": 6,
              "q_2ns": 3, "f_2ns": 3}
    order = ["cm1_n", "cm2_n", "cm3_n", "cm4_n", "tmpExp_n", "cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns", "q_2ns", "f_2ns"]
    off, o = {}, 0
    for s in order:
        off[s] = o
        o += widths[s] * (N if s.endswith("_n") else NE)
    deg = {s: (N if s.endswith("_n") else NE) for s in order}
    var = [{"section": "cm1_n", "dim": 1, "sectionPos": 0}, {"section": "cm1_n", "dim": 1, "sectionPos": 750}, {"section": "cm3_n", "dim": 3, "sectionPos": 405},
           {"section": "cm2_2ns", "dim": 1, "sectionPos": 7}, {"section": "q_2ns", "dim": 3, "sectionPos": 0}, {"section": "tmpExp_n", "dim": 3, "sectionPos": 12}]
    op = {"op": "mul", "dest": {"type": "tmp", "id": 4}, "src": [{"type": "cm", "id": 17, "prime": True}, {"type": "number", "value": "18446744069414584320"}]}
    step = {"tmpUsed": 9, "first": [op], "i": [], "last": []}
    j = {"starkStruct": {"nBits": 23, "nBitsExt": 24, "nQueries": 128, "verificationHashType": "GL", "steps": [{"nBits": b} for b in (24, 19, 14, 10, 6)]},
         "mapTotalN": o, "nConstants": 233, "nPublics": 44, "nCm1": 751, "nCm2": 168, "nCm3": 408, "nCm4": 6, "friExpId": 99, "nExps": 3000, "qDim": 3, "qDeg": 2,
         "mapDeg": deg, "mapOffsets": off, "mapSections": {s: list(range(3)) for s in order}, "mapSectionsN": widths,
         "mapSectionsN1": {s: widths[s] for s in order}, "mapSectionsN3": {s: 0 for s in order}, "varPolMap": var, "qs": [1, 2], "cm_n": [0, 1], "cm_2ns": [3],
         "peCtx": [{"tExpId": 1, "fExpId": 2, "zId": 3, "c1Id": 4, "numId": 5, "denId": 6, "c2Id": 7}],
         "puCtx": [{"tExpId": 1, "fExpId": 2, "h1Id": 8, "h2Id": 9, "zId": 3, "c1Id": 4, "numId": 5, "denId": 6, "c2Id": 7}] * 2,
         "ciCtx": [], "evMap": [{"type": "cm", "id": 5, "prime": False}, {"type": "const", "id": 2, "prime": True}, {"type": "q", "id": 0, "prime": False}],
         "step2prev": step, "step3prev": step, "step3": step, "step42ns": step, "step52ns": step,
         "exps_n": [None, 4], "q_2ns": [None], "cm4_n": [], "cm4_2ns": [1], "tmpExp_n": [None, None, 7], "exp2pol": {"12": 5, "40": 2}}
    return j, off, widths, var, N, NE


def test_starkinfo_mirror_loads_zkevm_geometry(tmp_path):
    """StarkInfo (stark_info.cpp:21-520) in the C++ mirror; no GPU involved.  The slab offsets must equal the ones
    recovered from the reference's generated zkEVM step tables (zkevm.chelpers.step*.parser.hpp arguments)."""
    j, off, widths, var, N, NE = _zkevm_like_starkinfo()
    # offsets found in the reference's tables (tests/expr_real.py clusters the same values)
    assert off["cm2_n"] == 6299844608 and off["cm3_n"] == 7709130752 and off["cm4_n"] == 11131682816 and off["tmpExp_n"] == 11182014464
    assert off["cm1_2ns"] == 14445182976 and off["cm2_2ns"] == 27044872192 and off["cm3_2ns"] == 29863444480 and off["cm4_2ns"] == 36708548608
    assert off["q_2ns"] == 36809211904 and off["f_2ns"] == 36859543552 and j["mapTotalN"] == 36909875200
    f = tmp_path / "zkevm.starkinfo.json"
    f.write_text(json.dumps(j, indent=1))
    r = subprocess.run([EXE, "starkinfo", str(f)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    assert (got["nBits"], got["nBitsExt"], got["nQueries"], got["hash"], got["steps"]) == (23, 24, 128, "GL", [24, 19, 14, 10, 6])
    assert got["mapTotalN"] == j["mapTotalN"] and got["nConstants"] == 233 and got["qDeg"] == 2
    assert got["constTreeBytes"] == (233 * NE + NE * 4 + (NE - 1) * 4 + 2) * 8
    sec_order = ["cm1_n", "cm1_2ns", "cm2_n", "cm2_2ns", "cm3_n", "cm3_2ns", "cm4_n", "cm4_2ns", "tmpExp_n", "q_2ns", "f_2ns"]     # eSection numbering
    assert got["offsets"] == [off[s] for s in sec_order]
    for v, p in zip(var, got["pols"]):
        rows = N if v["section"].endswith("_n") else NE
        base = off[v["section"]] + v["sectionPos"]
        assert p == [base, rows, v["dim"], widths[v["section"]], rows * v["dim"] * 8, base + 3 * widths[v["section"]]]
    assert got["evMap"] == [[0, 5, 0], [1, 2, 1], [2, 0, 0]]
    assert (got["puCtx"], got["peCtx"], got["ciCtx"], got["step3_ops"]) == (2, 1, 0, 1)
    assert got["step3_first_src"] == [[14, 17, 1, ""], [8, 0, 0, "18446744069414584320"]] and got["exp2pol"] == 2 and got["tmpExp_n"] == 3
    bad = tmp_path / "bad.json"; bad.write_text('{"starkStruct": {"nBits": 3}}')
    r = subprocess.run([EXE, "starkinfo", str(bad)], capture_output=True, text=True)
    assert r.returncode != 0 and "missing member" in r.stderr


def _joinzkin_expected(z1, z2, vk, steps):
    """plain restatement of joinzkin (src/starkpil/fri/proof2zkinStark.cpp:84-175) on Python dicts (insertion-ordered)"""
    out = {"publics": z1["publics"][:19] + z2["publics"][19:44]}
    fixed = ["publics", "root1", "root2", "root3", "root4", "evals", "s0_vals1", "s0_vals3", "s0_vals4", "s0_valsC",
             "s0_siblings1", "s0_siblings3", "s0_siblings4", "s0_siblingsC"]
    for pre, z in (("a_", z1), ("b_", z2)):
        for k in fixed:
            out[pre + k] = z[k]
        for i in range(1, steps):
            for suf in ("_root", "_siblings", "_vals"):
                out[f"{pre}s{i}{suf}"] = z[f"s{i}{suf}"]
        out[pre + "finalPol"] = z["finalPol"]
    out["rootC"] = [str(v) for v in vk["constRoot"]]
    return out


def test_joinzkin_mirror(tmp_path):
    """joinzkin of the C++ mirror (host/xlp_host.hpp) on two synthetic zkin objects with the golden proofs' member set:
    same members, same order, same values as the reference's function"""
    import random
    rng = random.Random(7)
    steps, nq = 4, 3
    fe = lambda: str(rng.randrange(0xFFFFFFFF00000001))

    def zkin():
        z = {"root1": [fe() for _ in range(4)], "root2": [fe() for _ in range(4)], "root3": [fe() for _ in range(4)], "root4": [fe() for _ in range(4)],
             "evals": [[fe(), fe(), fe()] for _ in range(5)]}
        for i in range(1, steps):
            z[f"s{i}_root"] = [fe() for _ in range(4)]
            z[f"s{i}_vals"] = [[fe() for _ in range(6)] for _ in range(nq)]
            z[f"s{i}_siblings"] = [[[fe() for _ in range(4)] for _ in range(3)] for _ in range(nq)]
        for t in ("1", "3", "4", "C"):
            z[f"s0_vals{t}"] = [[fe() for _ in range(2)] for _ in range(nq)]
            z[f"s0_siblings{t}"] = [[[fe() for _ in range(4)] for _ in range(5)] for _ in range(nq)]
        z["finalPol"] = [[fe(), fe(), fe()] for _ in range(4)]
        z["publics"] = [fe() for _ in range(44)]
        return z
    z1, z2 = zkin(), zkin()
    vk = {"constRoot": [rng.randrange(0xFFFFFFFF00000001) for _ in range(4)]}
    for name, obj in (("a.json", z1), ("b.json", z2), ("vk.json", vk)):
        (tmp_path / name).write_text(json.dumps(obj))
    r = subprocess.run([EXE, "joinzkin", str(tmp_path / "a.json"), str(tmp_path / "b.json"), str(tmp_path / "vk.json"), str(steps)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-500:]
    got = json.loads(r.stdout)
    exp = _joinzkin_expected(z1, z2, vk, steps)
    assert list(got.keys()) == list(exp.keys())
    assert got == exp


def _zkin_from_fixture(F):
    """a zkin object with the reference's member set made of a golden fixture's content (the queries the fixture keeps)"""
    z = {k: F[k] for k in ("root1", "root2", "root3", "root4", "evals")}
    nsteps = len(F["steps_nBits"])
    for i in range(1, nsteps):
        z[f"s{i}_root"] = F["fri_roots"][i - 1]
        z[f"s{i}_vals"] = [Q[f"s{i}_vals"] for Q in F["queries"]]
        z[f"s{i}_siblings"] = [Q[f"s{i}_siblings"] for Q in F["queries"]]
    for t in ("1", "3", "4", "C"):
        z[f"s0_vals{t}"] = [Q[f"s0_vals{t}"] for Q in F["queries"]]
        z[f"s0_siblings{t}"] = [Q[f"s0_siblings{t}"] for Q in F["queries"]]
    z["finalPol"] = F["finalPol"]
    z["publics"] = F["publics"][:44]                 # the zkin files hold 44; the fixture appends recursive2's constRoot (transcript input)
    return z


def _aggregate(tmp_path, za, zb, vk, steps):
    for name, obj in (("a.json", za), ("b.json", zb), ("vk.json", vk)):
        (tmp_path / name).write_text(json.dumps(obj))
    r = subprocess.run([EXE, "aggregate", str(tmp_path / "a.json"), str(tmp_path / "b.json"), str(tmp_path / "vk.json"), str(steps)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-500:]
    return json.loads(r.stdout)


def test_aggregation_chain_of_the_golden_proofs(tmp_path, golden_dir):
    """REFERENCE-HELD PIN of aggregation: the reference aggregated its four golden batch proofs pairwise and once more
    (recursive1.zkin.proof_{0..3} -> recursive2.zkin.proof_{01,23} -> proof_03).  Through the C++ mirror -- the input checks of
    Prover::genAggregatedProof (prover.cpp:695-728) and joinzkin (proof2zkinStark.cpp:84-175) -- every pair the reference joined
    is accepted, the stitched publics are the golden aggregated proof's publics, `rootC` is recursive2's constant root (which
    the golden recursive2 proofs open), and every member of the two inputs reappears under a_ / b_."""
    chain = json.load(open(os.path.join(golden_dir, "aggregation_chain.json")))
    fx = lambda n: json.load(open(os.path.join(golden_dir, n)))
    vk = {"constRoot": [int(x) for x in chain["recursive2_constRoot"]]}
    for J in chain["joins"]:
        A, B, OUT = fx(J["a"]), fx(J["b"]), fx(J["out"])
        za, zb = _zkin_from_fixture(A), _zkin_from_fixture(B)
        steps = len(A["steps_nBits"])
        got = _aggregate(tmp_path, za, zb, vk, steps)
        assert "result" not in got, got
        assert got["publics"] == OUT["publics"][:44]                      # the reference's own aggregated proof carries exactly these
        assert got["rootC"] == chain["recursive2_constRoot"] == OUT["constRoot"] == OUT["publics"][44:48]
        assert got == _joinzkin_expected(za, zb, vk, steps) and list(got.keys()) == list(_joinzkin_expected(za, zb, vk, steps).keys())
        for pre, z in (("a_", za), ("b_", zb)):
            for k, v in z.items():
                assert got[pre + k] == v
    # proofs that do not follow each other are refused with the reference's result code (ZKR_AGGREGATED_PROOF_INVALID_INPUT = 9)
    p = [_zkin_from_fixture(fx(f"proof_recursive1_{i}.json")) for i in range(4)]
    for a, b, what in ((1, 0, "newStateRoot"), (0, 2, "newStateRoot"), (0, 3, "newStateRoot"), (2, 1, "newStateRoot")):
        got = _aggregate(tmp_path, p[a], p[b], vk, 5)
        assert got["result"] == 9 and what in got["why"], got
    for idx, what in ((17, "chainId"), (18, "forkId"), (43, "BatchNum"), (30, "AccInputHash")):
        bad = dict(p[0]); bad["publics"] = list(bad["publics"]); bad["publics"][idx] = str(int(bad["publics"][idx]) + 1)
        got = _aggregate(tmp_path, bad, p[1], vk, 5)
        assert got["result"] == 9 and what in got["why"], got


@pytest.mark.gpu
def test_host_boundary_is_reentrant_from_openmp_threads(orc):
    """FRIProve::prove calls NTT_Goldilocks(nX, 1).INTT from inside an OpenMP region (friProve.cpp:44-108).  The C++ mirror's
    INTT (+ a transcript-style hash_full_result) from 8 threads at once over 192 groups: every result equals the oracle's."""
    nX, groups, seed = 32, 192, 0x5EED00E1
    env = dict(os.environ, OMP_NUM_THREADS="8")
    r = subprocess.run([EXE, "omp_intt", str(nX), str(groups), str(seed)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    got = json.loads(r.stdout)
    assert got["threads"] == 8
    for g in range(groups):
        src = orc.fill(nX * 3, seed, start=g * nX * 3).reshape(nX, 3)
        assert (np.array(got["intt"][g], np.uint64).reshape(nX, 3) == orc.ntt(src, nX, 3, inverse=True)).all(), g
        st = orc.fill(12, seed + 1, start=g * 12)
        assert got["perm"][g] == [int(v) for v in orc.perm(st)], g
