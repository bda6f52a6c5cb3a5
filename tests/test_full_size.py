"""BASELINE.json configs[2] and configs[3] AT SIZE, checked against the CPU oracle through size-independent properties:

  cfg3  Poseidon-Goldilocks Merkle commit of a 2^24-row extended trace, 751 columns (the real cm1 width) and 670 columns
        (the width BASELINE's text quotes): PoseidonGoldilocks::merkletree_avx (merkleTreeGL.cpp:37-44), 100.8 / 89.9 GB of rows
  cfg4  Starks::genProof (starks.cpp:9-404) at the fork-9 zkEVM geometry, N = 2^23: see test_zkevm_genproof_2p23_checked

Bit-exact: integer field."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def xlp():
    from xlayer_prover_amd import capi
    capi.init(0)
    return capi


@pytest.mark.parametrize("w", [751, 670])
def test_merkle_config3_full_size(xlp, orc, w):
    """2^24 x w rows resident in HBM (dev_fill: the synthetic trace of SURVEY 8d, regenerated row by row on the oracle side):
    (a) root == cap tree (oracle permutations) over the roots of the 8 sub-trees of 2^21 rows each, built by separate launches
        -- the perfect-binary-tree property the 8-GPU cap assembly of SURVEY 8e relies on;
    (b) 24 sampled leaves == orc.linear_hash of the same rows (incl. first / last row, sub-tree boundaries);
    (c) their sibling paths, read from the node buffer, verify against the root with the oracle;
    (d) xlp_dev_merkle_proofs (the batched getGroupProof) returns the same rows and paths."""
    import torch
    h = 1 << 24
    seed = 0x5EED0003
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < (h * w + 3 * h * 4) * 8 + (8 << 30):
        pytest.skip("not enough free HBM for the configs[2] section")
    src = torch.empty(h * w, dtype=torch.int64, device="cuda")
    xlp.dev_fill(src, seed, 0)
    nodes = torch.empty((2 * h - 1) * 4, dtype=torch.int64, device="cuda")
    xlp.dev_merkle(nodes, src, w, h)
    xlp.sync()
    root = xlp.to_np(nodes[-4:])
    # (a)
    G, q = 8, h // 8
    sub = torch.empty((2 * q - 1) * 4, dtype=torch.int64, device="cuda")
    caps = []
    for g in range(G):
        xlp.dev_merkle(sub, src[g * q * w:(g + 1) * q * w], w, q)
        xlp.sync()
        caps.append(xlp.to_np(sub[-4:]).copy())
    lvl = caps
    while len(lvl) > 1:
        lvl = [orc.perm(list(lvl[2 * i]) + list(lvl[2 * i + 1]) + [0] * 4)[:4] for i in range(len(lvl) // 2)]
    assert [int(x) for x in lvl[0]] == [int(x) for x in root]
    # the sub-roots are also where the perfect tree puts them: level log2(q), index g
    off, cnt = 0, h
    level_off = []
    while cnt >= 1:
        level_off.append((off, cnt)); off += 4 * cnt; cnt //= 2
    lo, lc = level_off[21]
    assert lc == G
    got_caps = xlp.to_np(nodes[lo:lo + 4 * G]).reshape(G, 4)
    assert (got_caps == np.array(caps)).all()
    # (b) + (c): pull only the nodes each path needs
    rng = np.random.default_rng(w)
    idxs = [0, 1, h - 1, q - 1, q, 5 * q + 12345] + [int(x) for x in rng.integers(0, h, 18)]
    for idx in idxs:
        row = orc.fill(w, seed, idx * w)
        assert (xlp.to_np(src[idx * w:(idx + 1) * w]) == row).all()
        leaf = xlp.to_np(nodes[4 * idx:4 * idx + 4])
        assert [int(x) for x in leaf] == [int(x) for x in orc.linear_hash(row)], idx
        sib, i = [], idx
        for o, c in level_off[:-1]:
            s = (i ^ 1) * 4
            sib.append(xlp.to_np(nodes[o + s:o + s + 4])); i >>= 1
        assert [int(x) for x in orc.merkle_root_from_path(row, np.concatenate(sib), idx)] == [int(x) for x in root], idx
    # (d)
    proofs = xlp.dev_merkle_proofs(src, nodes, w, h, idxs).reshape(len(idxs), -1)
    assert proofs.shape[1] == w + 4 * 24
    for p, idx in zip(proofs, idxs):
        assert (p[:w] == orc.fill(w, seed, idx * w)).all()
        assert [int(x) for x in orc.merkle_root_from_path(p[:w], p[w:], idx)] == [int(x) for x in root]
    del src, nodes, sub
    torch.cuda.empty_cache()


def _checked_run(xlp, args, timeout):
    """tests/zkevm_genproof_checked.py in its own process (this one first gives its HBM back: the 2^23 run needs ~293 GB)"""
    import torch
    torch.cuda.empty_cache()
    xlp.shutdown()
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "zkevm_genproof_checked.py")] + args, capture_output=True, text=True, timeout=timeout)
    finally:
        xlp.init(0)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


_CHECKED_2P23 = {}


def _checked_2p23(xlp):
    """the 2^23 checked run, ONCE per session (it takes ~100 s; two tests read it): the Python orchestration of genProof on ROW-MAJOR sections (the
    reference's layout), its step programs interpreted, checked against the oracle stage by stage (16 row windows per program)"""
    if "out" not in _CHECKED_2P23:
        _CHECKED_2P23["out"] = _checked_run(xlp, ["--circuit", "zkevm", "--log-n", "23"], 1100)
    return _CHECKED_2P23["out"]


def _programs_built(circ):
    from xlayer_prover_amd import programs as R
    return os.path.exists(os.path.join(R.REF_DIR, f"{circ}.step52ns.xlpexpr"))


@pytest.mark.parametrize("circ,args", [("recursive1", []), ("c12a", []), ("zkevm", ["--log-n", "14", "--aliased", "1"]), ("zkevm", ["--log-n", "14", "--aliased", "0"])])
def test_genproof_small_geometries_checked(xlp, circ, args):
    """the checked run on the two small STARKs of genBatchProof at their real geometry (prover.cpp:610,644) and on the zkEVM
    programs at reduced row counts, both slab layouts: every check of tests/zkevm_checks.py"""
    if not _programs_built(circ):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    out = _checked_run(xlp, ["--circuit", circ] + args, 900)
    c = out["checks"]
    assert c["transcript_replayed"] and c["evals_checked"] >= 8 and c["fri_folds"] > 0 and c["merkle_paths"] > 0
    assert all(v >= 64 for v in c["step_program_rows"].values()) and len(c["step_program_rows"]) >= 4


def test_zkevm_genproof_2p23_checked(xlp):
    """BASELINE configs[3] at the real size on one GPU: Starks::genProof at the fork-9 zkEVM geometry, N = 2^23 (aliased 239 GB
    slab), with the step-program forms that bench.py times (dot-product rewrite + ahead-of-time plugins, FRI polynomial as
    table program + xlp_dev_dot_columns), CHECKED against the CPU oracle: for each of the five programs >= 64 sampled rows equal
    the as-converted reference program run on the oracle; >= 8 evaluations recomputed from the resident columns; all
    128 x 5 Merkle openings and every FRI tree path verify; the fold chain of all 128 queries down to the final polynomial; the
    transcript replays.  The checked pass has the same roots as the timed pass."""
    import torch
    if not _programs_built("zkevm"):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    _, total = torch.cuda.mem_get_info()
    if total < 300e9:
        pytest.skip("needs the 288 GB of an MI355X")
    out = _checked_2p23(xlp)
    c = out["checks"]
    assert set(c["step_program_rows"]) == {"step2prev", "step3prev", "step3", "step42ns", "step52ns"}
    assert all(v >= 64 for v in c["step_program_rows"].values())
    assert c["evals_checked"] >= 8 and c["transcript_replayed"]
    assert c["merkle_paths"] == 128 * 5 + 128 * 4 and c["fri_folds"] == 128 * 4 and c["fri_first_tree_vs_f"] == 128
    assert "dot_columns" in out["programs"]["step52ns"]          # (the row-major step programs run interpreted since round 5: no row-major plugin is built)


def test_cpp_starks_2p23_equals_checked_python_flow(xlp):
    """the C++ Starks::genProof (host/starks.hpp, reference-shaped entry point; witness in page-locked host memory, upload pipelined with
    step 1; COLUMN-MAJOR device sections and ahead-of-time step programs: the product path) at the fork-9 zkEVM geometry, N = 2^23, from the
    bundle build() writes: its zkin proof has the roots, FRI roots and
    final polynomial of the Python orchestration that test_zkevm_genproof_2p23_checked checks against the oracle stage by stage,
    and its openings verify against those roots on the oracle"""
    import torch
    from xlayer_prover_amd import programs as R
    from tests.zkin_verify import verify_zkin, ints
    from oracle import gl_oracle as orc
    bundle = os.path.join(R.REF_DIR, "bundles", "zkevm.geo23ac")       # the product's bundle: column-major sections, ahead-of-time kernels
    if not os.path.exists(os.path.join(bundle, "zkevm.steps.json")):
        pytest.skip("zkEVM bundle not built (needs the reference tree at build time)")
    _, total = torch.cuda.mem_get_info()
    if total < 300e9:
        pytest.skip("needs the 288 GB of an MI355X")
    exp = _checked_2p23(xlp)
    torch.cuda.empty_cache()
    xlp.shutdown()
    try:
        exe = os.path.join(ROOT, "xlayer_prover_amd", "host", "test_host_shim")
        r = subprocess.run([exe, "bundle", bundle, "zkevm", str(0x5EED0401), str(0x5EED0402)], capture_output=True, text=True, timeout=900)
    finally:
        xlp.init(0)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stderr.count("compiled") >= 4 and "dot_columns" in r.stderr
    Z = json.loads(r.stdout)
    assert [ints(Z[k]) for k in ("root1", "root2", "root3", "root4")] == exp["roots"]
    fr = exp["fri_roots_and_head_of_final_pol"]
    steps = [24, 19, 14, 10, 6]
    for k in range(1, len(steps)):
        assert ints(Z[f"s{k}_root"]) == fr[k - 1]
    assert [x for e in Z["finalPol"] for x in ints(e)][:6] == fr[-1]
    rep = verify_zkin(orc, Z, 23, 24, steps, 128, {"1": 751, "2": 168, "3": 408, "4": 6, "C": 234})
    assert rep["folds"] == 128 * 4 and rep["paths"] == 128 * 9
