"""Slab geometry of the three STARKs of Prover::genBatchProof (tools/zkevm_geometry.py): the offsets recovered from the
reference's generated step programs, the relocation of the zkEVM programs and the one-GPU aliased slab whose
sections share addresses only when their lifetimes are disjoint.  CPU only; the GPU run of the layouts is
tools/bench_zkevm_genproof.py (identical roots for both layouts)."""
import os

import pytest

from tests import expr_real as R
from tools import zkevm_geometry as G
from xlayer_prover_amd import expr as E

needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(R.REF_DIR, "zkevm.step42ns.xlpexpr")),
                               reason="converted reference programs not built (needs the reference tree at build time)")


def test_real_zkevm_offsets_follow_the_packing_rule():
    G._check_real()
    L = G.layout(23)
    assert L["end"][0] * 8 == 295279001600                       # the reference's slab: 295 GB
    A = G.layout(23, aliased=True)
    assert A["end"][0] * 8 < 240e9 and A["end"][0] == 3560 << 23


def _span(L, name):
    o, w, rows = L[name]
    return o, o + w * rows


def test_aliased_sections_overlap_only_dead_ones():
    """what is alive when (starks.cpp:9-404 + the step programs' reads, tests below): base-domain sections until the last
    base-domain program (step3); cm3_n until its extension; extended sections from their LDE to the end."""
    for ln in (10, 14, 23):
        A = G.layout(ln, aliased=True)
        disjoint = lambda a, b: _span(A, a)[1] <= _span(A, b)[0] or _span(A, b)[1] <= _span(A, a)[0]
        inside = lambda a, lo, hi: lo <= _span(A, a)[0] and _span(A, a)[1] <= hi
        base = ["cm1_n", "cm2_n", "cm3_n", "cm4_n", "tmpExp_n"]
        early_ext = ["cm1_2ns", "cm2_2ns"]                        # written while the base-domain sections are alive
        for a in base + early_ext:
            for b in base + early_ext:
                assert a == b or disjoint(a, b), (a, b)
        # cm3_2ns is written by the step-3 LDE, which reads cm3_n: only cm1_n / cm2_n may lie under it
        assert inside("cm3_2ns", _span(A, "cm1_n")[0], _span(A, "cm2_n")[1]) and disjoint("cm3_2ns", "cm3_n")
        # from step 4 on: everything extended + the re-extended constants + q / f, pairwise disjoint
        late = ["cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns", "q_2ns", "f_2ns", "const_2ns"]
        for a in late:
            for b in late:
                assert a == b or disjoint(a, b), (a, b)
        # the constants go where cm3_n .. tmpExp_n were (dead after the step-3 LDE), never under a live extended section
        assert inside("const_2ns", _span(A, "cm3_n")[0], _span(A, "tmpExp_n")[1])
        assert all(inside(x, _span(A, "tmpExp_n")[0], _span(A, "tmpExp_n")[1]) for x in ("cm4_2ns", "q_2ns", "f_2ns"))
        assert A["end"][0] == _span(A, "cm2_2ns")[1]


def _sections_touched(code, L, secs):
    rd, wr = set(), set()
    names = [n for n, w, _ in secs if w]
    for i, (_, _, kind, dim, q) in enumerate(R.operands(code)):
        if kind not in (E.K_POLS, E.K_POLS_PRIME):
            continue
        off, stride = int(code[q]), int(code[q + 1] if kind == E.K_POLS else code[q + 3])
        hit = [n for n in names if L[n][1] == stride and L[n][0] <= off < L[n][0] + stride]
        assert len(hit) >= 1, (off, stride)
        (wr if i % 3 == 0 else rd).update(hit[:1] if len(hit) == 1 else hit)
    return rd, wr


@needs_ref
@pytest.mark.parametrize("aliased", [False, True])
def test_zkevm_programs_respect_the_section_lifetimes(aliased):
    """the relocated programs only touch sections that are alive at their stage -- the premise of the aliased slab"""
    ln = 12
    progs, info = G.load(ln, aliased)
    L = G.layout(ln, aliased)
    assert max(info["n_const"], info["n_const2"]) == G.N_CONST and info["nev"] == 1973
    base = {"cm1_n", "cm2_n", "cm3_n", "cm4_n", "tmpExp_n"}
    ext = {"cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns"}
    for st, (code, n1, n3) in progs.items():
        if aliased:          # overlapping sections with equal widths would be ambiguous: resolve by the domain of the step
            names = [(n, w, e) for n, w, e in G.SECTIONS if (n in ext) == st.endswith("2ns")]
        else:
            names = G.SECTIONS
        rd, wr = _sections_touched(code, L, names)
        if st.endswith("2ns"):
            assert rd <= ext and not wr, (st, rd, wr)             # q_2ns / f_2ns are written through their own operands
        else:
            assert rd <= base and wr <= {"cm3_n", "tmpExp_n"}, (st, rd, wr)


@needs_ref
@pytest.mark.parametrize("circ", ["c12a", "recursive1", "recursive2"])
def test_small_starks_keep_their_real_geometry(circ):
    """every column operand of the converted c12a / recursive1 programs falls inside the section its stride names
    (load() asserts it), and the evaluation counts are the golden proofs' (recursive1: evals [118][3])"""
    C = G.CIRCUITS[circ]
    progs, info = G.load(C["log_n"], False, circ)
    assert set(progs) == set(C["steps"])
    assert info["nev"] == {"c12a": 146, "recursive1": 118, "recursive2": 118}[circ] and info["n_const2"] == 52
    L = G.layout(C["log_n"], False, circ)
    assert L["cm1_2ns"][2] == 1 << (C["log_n"] + C["ext"])
    tab, dots, row, n_ext = G.load_fri_dots(C["log_n"], False, circ)
    assert dots["ngroups"] == 3 and row.ninstr < 20


def _run_bench(*args):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_zkevm_genproof.py"), "--no-jit", *args], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-1500:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@needs_ref
@pytest.mark.gpu
def test_aliased_slab_gives_the_reference_layouts_proof():
    """every stage of genProof at the zkEVM geometry (2^12 rows, the five converted reference programs) on the one-GPU slab
    (cm3_2ns over cm1_n / cm2_n, const_2ns re-extended over cm3_n, ...) == the same run on the reference's slab layout:
    the four roots, every FRI root and the final polynomial"""
    a = _run_bench("--log-n", "12", "--aliased", "0")
    b = _run_bench("--log-n", "12", "--aliased", "1")
    assert a["roots"] == b["roots"] and a["fri_roots_and_head_of_final_pol"] == b["fri_roots_and_head_of_final_pol"]
    assert b["layout"].startswith("one-GPU") and "STARK_STEP_4_CONST_LDE" in b["timers_s"]


@needs_ref
@pytest.mark.gpu
def test_recursive1_geometry_run_commits_the_golden_empty_tree(golden_dir):
    """recursive1 has no lookups: its cm2 tree is the tree over 2^20 zero-width rows, and the genProof run at recursive1's real
    geometry (converted reference programs, golden FRI steps / queries) must commit exactly the root2 of the reference's
    recursive1 proof (testvectors/aggregatedProof/recursive1.zkin.proof_0.json, fixture tests/golden/proof_recursive1_0.json)"""
    import json
    g = json.load(open(os.path.join(golden_dir, "proof_recursive1_0.json")))
    o = _run_bench("--circuit", "recursive1")
    assert [str(v) for v in o["roots"][1]] == g["root2"]
    assert len(o["fri_roots_and_head_of_final_pol"]) == len(g["steps_nBits"])
