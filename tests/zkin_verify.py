"""Verifier-side checks of a complete zkin proof (the JSON proof2zkinStark emits, src/starkpil/fri/proof2zkinStark.cpp:8-82) on the
CPU oracle -- what tests/test_oracle_golden.py does for the reference's two golden proofs, for proofs produced here:
transcript replay (challenges, FRI challenges, query indices), every Merkle path of the five step-0 trees and of the FRI trees,
every fold down to the final polynomial, and the final polynomial's degree.  TEST INFRASTRUCTURE."""
import numpy as np


def ints(x):
    if isinstance(x, list):
        return [ints(y) for y in x]
    return int(x)


def verify_zkin(orc, Z, n_bits, n_bits_ext, steps, n_queries, widths):
    """Z: parsed zkin (+ "publics", "verkey"); widths: {"1": w, "2": w, "3": w, "4": w, "C": w} of the five trees.
    Returns {"challenges", "ys", "special_x"}; raises AssertionError on any inconsistency."""
    t = orc.Transcript()
    t.put(ints(Z["verkey"])); t.put(ints(Z["publics"]))
    ch = {}
    t.put(ints(Z["root1"])); ch[0] = t.get_field(); ch[1] = t.get_field()
    t.put(ints(Z["root2"])); ch[2] = t.get_field(); ch[3] = t.get_field()
    t.put(ints(Z["root3"])); ch[4] = t.get_field()
    t.put(ints(Z["root4"])); ch[7] = t.get_field()
    for e in ints(Z["evals"]):
        t.put(e)
    ch[5] = t.get_field(); ch[6] = t.get_field()
    special = []
    for si in range(len(steps)):
        special.append(t.get_field())
        if si < len(steps) - 1:
            t.put(ints(Z[f"s{si + 1}_root"]))
        else:
            for e in ints(Z["finalPol"]):
                t.put(e)
    ys = [int(y) for y in t.get_permutations(n_queries, steps[0])]
    roots0 = {"1": Z["root1"], "2": Z["root2"], "3": Z["root3"], "4": Z["root4"], "C": Z["verkey"]}
    checked = {"paths": 0, "folds": 0}
    for q, idx in enumerate(ys):
        for tn, root in roots0.items():
            if widths[tn] == 0:
                assert f"s0_vals{tn}" not in Z                 # absent sections are omitted (proof2zkinStark.cpp)
                continue
            row = ints(Z[f"s0_vals{tn}"][q])
            assert len(row) == widths[tn]
            r = orc.merkle_root_from_path(row, ints(Z[f"s0_siblings{tn}"][q]), idx)
            assert [int(x) for x in r] == ints(root), (tn, q)
            checked["paths"] += 1
        sinv = orc.inv(7)
        for si in range(1, len(steps)):
            g = idx % (1 << steps[si])
            vals = ints(Z[f"s{si}_vals"][q])
            r = orc.merkle_root_from_path(vals, ints(Z[f"s{si}_siblings"][q]), g)
            assert [int(x) for x in r] == ints(Z[f"s{si}_root"]), ("fri", si, q)
            checked["paths"] += 1
            bits = steps[si - 1]
            got = orc.fri_fold_one(vals, bits, bits - steps[si], g, special[si], sinv)
            if si < len(steps) - 1:
                nb = steps[si + 1]
                exp = ints(Z[f"s{si + 1}_vals"][q])[3 * (g >> nb): 3 * (g >> nb) + 3]
            else:
                exp = ints(Z["finalPol"][g])
            assert [int(x) for x in got] == exp, ("fold", si, q)
            checked["folds"] += 1
            for _ in range(bits - steps[si]):
                sinv = orc.mul(sinv, sinv)
    fp = np.array(ints(Z["finalPol"]), np.uint64)
    co = orc.ntt(fp, fp.shape[0], 3, inverse=True)
    assert not co[fp.shape[0] >> (n_bits_ext - n_bits):].any(), "final polynomial has too high a degree"
    return {"challenges": {k: [int(x) for x in v] for k, v in ch.items()}, "ys": ys, "special_x": [[int(x) for x in s] for s in special], **checked}


def verify_zkin_bn128(orc, bo, K, Z, n_bits, n_bits_ext, steps, n_queries, widths):
    """the same checks for a recursivef-style zkin (host/starkRecursiveF.hpp proof2zkinStarkC12): TranscriptBN128 replay (first input:
    the constant root, starkRecursiveF.cpp:171-174), MerkleTreeBN128 openings ([level][16] sibling groups), FRI folds, final degree.
    Z["verkey"] may be absent: it is then rebuilt from the first opening of the constant tree (the position inside each 16-node
    group is found by matching), and the replayed first query index must be the position that walk discovered."""
    def root_from_groups(row, groups, idx=None):
        cur, pos, sh = bo.linear_hash_row(K, row), 0, 0
        for g in groups:
            g = [int(x) for x in g]
            if idx is None:
                assert g.count(cur) == 1
                p = g.index(cur)
            else:
                p = (idx >> sh) & 15
                assert g[p] == cur
            pos |= p << sh; sh += 4
            cur = bo.poseidon(K, [0] + g)[0]
        return cur, pos
    if "verkey" not in Z:
        vk, idx0 = root_from_groups(ints(Z["s0_valsC"][0]), Z["s0_siblingsC"][0])
    else:
        vk, idx0 = int(Z["verkey"]), None
    t = bo.TranscriptBN128(K)
    t.put([vk]); t.put(ints(Z["publics"]))
    ch = {}
    t.put([int(Z["root1"])]); ch[0] = t.get_field(); ch[1] = t.get_field()
    t.put([int(Z["root2"])]); ch[2] = t.get_field(); ch[3] = t.get_field()
    t.put([int(Z["root3"])]); ch[4] = t.get_field()
    t.put([int(Z["root4"])]); ch[7] = t.get_field()
    for e in ints(Z["evals"]):
        t.put(e)
    ch[5] = t.get_field(); ch[6] = t.get_field()
    special = []
    for si in range(len(steps)):
        special.append(t.get_field())
        if si < len(steps) - 1:
            t.put([int(Z[f"s{si + 1}_root"])])
        else:
            for e in ints(Z["finalPol"]):
                t.put(e)
    ys = [int(y) for y in t.get_permutations(n_queries, steps[0])]
    if idx0 is not None:
        assert ys[0] == idx0, "the first query index is not where the constant-tree opening sits"
    roots0 = {"1": int(Z["root1"]), "2": int(Z["root2"]), "3": int(Z["root3"]), "4": int(Z["root4"]), "C": vk}
    checked = {"paths": 0, "folds": 0}
    for q, idx in enumerate(ys):
        for tn, root in roots0.items():
            if widths[tn] == 0 and f"s0_vals{tn}" not in Z:
                continue
            row = ints(Z[f"s0_vals{tn}"][q])
            assert len(row) == widths[tn]
            assert root_from_groups(row, Z[f"s0_siblings{tn}"][q], idx)[0] == root, (tn, q)
            checked["paths"] += 1
        sinv = orc.inv(7)
        for si in range(1, len(steps)):
            g = idx % (1 << steps[si])
            vals = ints(Z[f"s{si}_vals"][q])
            assert root_from_groups(vals, Z[f"s{si}_siblings"][q], g)[0] == int(Z[f"s{si}_root"]), ("fri", si, q)
            checked["paths"] += 1
            bits = steps[si - 1]
            got = orc.fri_fold_one(vals, bits, bits - steps[si], g, [x % 0xFFFFFFFF00000001 for x in special[si]], sinv)
            if si < len(steps) - 1:
                nb = steps[si + 1]
                exp = ints(Z[f"s{si + 1}_vals"][q])[3 * (g >> nb): 3 * (g >> nb) + 3]
            else:
                exp = ints(Z["finalPol"][g])
            assert [int(x) for x in got] == exp, ("fold", si, q)
            checked["folds"] += 1
            for _ in range(bits - steps[si]):
                sinv = orc.mul(sinv, sinv)
    fp = np.array(ints(Z["finalPol"]), np.uint64)
    co = orc.ntt(fp, fp.shape[0], 3, inverse=True)
    assert not co[fp.shape[0] >> (n_bits_ext - n_bits):].any(), "final polynomial has too high a degree"
    return {"verkey": vk, "ys": ys, **checked}
