#!/usr/bin/env python3
"""Regenerate tests/golden/*.json from the reference's own test vectors.

Runs in the BUILD CONTAINER ONLY (needs /root/reference); the emitted fixtures
are data (inputs + expected outputs found in the reference's testvectors) and
are committed.  Nothing at test/bench run time reads /root/reference.

What it does:
 1. Poseidon KATs: every `db` entry with 12 hex words in testvectors/**/*.json
    satisfies key == perm(words)[0..3] (key string = h3|h2|h1|h0, 16 hex each;
    reference rule src/hashdb/smt.cpp:1070-1073, src/utils/scalar.cpp:101-108).
    All are checked against the oracle here; a sample is written out.
 2. Golden STARK proofs (outputs of Starks::genProof via proof2zkinStark):
    recover query index 0 by DFS on one Merkle path, derive the constant-tree
    root (= verkey), replay the Fiat-Shamir transcript, re-derive all query
    indices, check EVERY Merkle path and EVERY FRI fold with the oracle, then
    write a compact fixture (a few queries) for the test-suite.
"""
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import gl_oracle as orc  # noqa: E402

REF = "/root/reference/testvectors"
N_KAT_OUT = 128
QUERIES_OUT = [0, 1, 17, 42]


def poseidon_kats():
    kats = []
    for f in sorted(glob.glob(REF + "/**/*.json", recursive=True)):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if not isinstance(j, dict) or not isinstance(j.get("db"), dict):
            continue
        for key, val in j["db"].items():
            if not (isinstance(val, list) and len(val) == 12):
                continue
            words = [int(v, 16) for v in val]
            k = key[2:] if key.startswith("0x") else key
            k = k.rjust(64, "0")
            h = [int(k[48:64], 16), int(k[32:48], 16), int(k[16:32], 16), int(k[0:16], 16)]
            kats.append((words, h))
    # dedupe, keep order
    seen, uniq = set(), []
    for wds, h in kats:
        t = tuple(wds)
        if t not in seen:
            seen.add(t)
            uniq.append((wds, h))
    bad = 0
    for wds, h in kats:
        out = orc.perm(wds)
        if [int(x) for x in out[:4]] != h:
            bad += 1
    print(f"poseidon db KATs: {len(kats)} entries ({len(uniq)} unique), {bad} mismatches")
    assert bad == 0 and len(kats) >= 1243
    step = max(1, len(uniq) // N_KAT_OUT)
    sample = uniq[::step][:N_KAT_OUT]
    zero = [int(x) for x in orc.perm([0] * 12)]
    return {
        "source": "testvectors/**/*.json `db` objects (12-word values); key = perm(value)[0..3]",
        "total_checked": len(kats),
        "perm_zero": [str(x) for x in zero],
        "kats": [{"in": [str(x) for x in wds], "out4": [str(x) for x in h]} for wds, h in sample],
    }


def ints(x):
    if isinstance(x, list):
        return [ints(y) for y in x]
    return int(x)


def replay(J, verkey, extra_publics, steps_nbits, nqueries):
    """Transcript order of Starks::genProof (starks.cpp:28-29,60,68-69,141,150-151,
    222,234,293,306,336-342) + FRIProve::prove (friProve.cpp:29-30,125,130-134)."""
    t = orc.Transcript()
    t.put(verkey)
    t.put(ints(J["publics"]) + list(extra_publics))
    ch = {}
    t.put(ints(J["root1"])); ch[0] = t.get_field(); ch[1] = t.get_field()
    t.put(ints(J["root2"])); ch[2] = t.get_field(); ch[3] = t.get_field()
    t.put(ints(J["root3"])); ch[4] = t.get_field()
    t.put(ints(J["root4"])); ch[7] = t.get_field()
    for e in ints(J["evals"]):
        t.put(e)
    ch[5] = t.get_field(); ch[6] = t.get_field()
    special = []
    for si in range(len(steps_nbits)):
        special.append(t.get_field())
        if si < len(steps_nbits) - 1:
            t.put(ints(J[f"s{si+1}_root"]))
        else:
            for e in ints(J["finalPol"]):
                t.put(e)
    ys = t.get_permutations(nqueries, steps_nbits[0])
    return ch, special, ys


def golden_proof(path, extra_publics, name, queries_out=None):
    queries_out = QUERIES_OUT if queries_out is None else queries_out
    J = json.load(open(path))
    nq = len(J["s0_vals1"])
    nbits_ext = len(J["s0_siblings1"][0])
    steps = [nbits_ext] + [len(J[f"s{i}_siblings"][0]) for i in range(1, 5)]
    assert steps == [20, 16, 12, 9, 6], steps
    root1 = ints(J["root1"])
    idx0 = orc.find_merkle_index(ints(J["s0_vals1"][0]), ints(J["s0_siblings1"][0]), root1)
    assert idx0 != 2**64 - 1
    const_root = [int(x) for x in orc.merkle_root_from_path(ints(J["s0_valsC"][0]), ints(J["s0_siblingsC"][0]), idx0)]
    verkey = const_root
    if extra_publics is None:          # a recursive2 proof: publics[44..47] = its own constRoot
        extra_publics = const_root
    ch, special, ys = replay(J, verkey, extra_publics, steps, nq)
    ys = [int(y) for y in ys]
    assert ys[0] == idx0, (ys[0], idx0)
    print(f"{name}: idx0={idx0} constRoot={const_root}")

    # ---- every Merkle path
    trees0 = {"1": "root1", "3": "root3", "4": "root4"}
    nfail = 0
    for q in range(nq):
        for tname, rname in trees0.items():
            r = orc.merkle_root_from_path(ints(J[f"s0_vals{tname}"][q]), ints(J[f"s0_siblings{tname}"][q]), ys[q])
            nfail += [int(x) for x in r] != ints(J[rname])
        r = orc.merkle_root_from_path(ints(J["s0_valsC"][q]), ints(J["s0_siblingsC"][q]), ys[q])
        nfail += [int(x) for x in r] != const_root
        for si in range(1, 5):
            r = orc.merkle_root_from_path(ints(J[f"s{si}_vals"][q]), ints(J[f"s{si}_siblings"][q]), ys[q] % (1 << steps[si]))
            nfail += [int(x) for x in r] != ints(J[f"s{si}_root"])
    print(f"{name}: merkle path failures = {nfail} of {nq*8}")
    assert nfail == 0
    # empty tree (cm2 has width 0): root2 = 20-fold H(0|0)
    z = np.zeros(4, np.uint64)
    for _ in range(nbits_ext):
        z = orc.perm(list(z) + list(z) + [0] * 4)[:4]
    assert [int(x) for x in z] == ints(J["root2"])

    # ---- every FRI fold: step si>=1 folds tree-si rows into tree-(si+1) values / finalPol
    sinv = orc.inv(orc.shift())
    sinvs = [None, sinv]
    for si in range(1, 5):
        red = steps[si - 1] - steps[si]
        s = sinvs[si]
        for _ in range(red):
            s = orc.mul(s, s)
        sinvs.append(s)
    ffail = 0
    folds = []
    for q in range(nq):
        for si in range(1, 5):
            bits = steps[si - 1]
            red = bits - steps[si]
            g = ys[q] % (1 << steps[si])
            got = orc.fri_fold_one(ints(J[f"s{si}_vals"][q]), bits, red, g, special[si], sinvs[si])
            if si < 4:
                nb = steps[si + 1]
                exp = ints(J[f"s{si+1}_vals"][q])[3 * (g >> nb): 3 * (g >> nb) + 3]
            else:
                exp = ints(J["finalPol"][g])
            ok = [int(x) for x in got] == exp
            ffail += not ok
    print(f"{name}: FRI fold failures = {ffail} of {nq*4}")
    assert ffail == 0
    # step 0 is the copy (red = 0): special[0] is drawn but unused (friProve.cpp:29-30,87-90)

    fx = {
        "source": os.path.relpath(path, "/root/reference"),
        "nBits": 17, "nBitsExt": nbits_ext, "nQueries": nq, "steps_nBits": steps,
        "verkey": [str(x) for x in verkey],
        "publics": [str(x) for x in ints(J["publics"]) + list(extra_publics)],
        "root1": J["root1"], "root2": J["root2"], "root3": J["root3"], "root4": J["root4"],
        "constRoot": [str(x) for x in const_root],
        "evals": J["evals"], "finalPol": J["finalPol"],
        "fri_roots": [J[f"s{i}_root"] for i in range(1, 5)],
        "challenges": {str(k): [str(int(x)) for x in v] for k, v in ch.items()},
        "special_x": [[str(int(x)) for x in s] for s in special],
        "shift_inv_per_step": [str(s) if s is not None else None for s in sinvs[:5]],
        "ys": ys,
        "queries": [],
    }
    for q in queries_out:
        e = {"q": q, "idx": ys[q]}
        for tname in ["1", "3", "4", "C"]:
            e[f"s0_vals{tname}"] = J[f"s0_vals{tname}"][q]
            e[f"s0_siblings{tname}"] = J[f"s0_siblings{tname}"][q]
        for si in range(1, 5):
            e[f"s{si}_vals"] = J[f"s{si}_vals"][q]
            e[f"s{si}_siblings"] = J[f"s{si}_siblings"][q]
        fx["queries"].append(e)
    return fx, const_root


# The seven golden proofs of the reference's e2e flow (test/service/aggregator/aggregator_service.cpp:31-144): four batch proofs
# (recursive1.zkin.proof_{0..3} = genBatchProof outputs), aggregated pairwise (recursive2.zkin.proof_{01,23} = genAggregatedProof
# outputs) and once more (proof_03).  (fixture name, source, is_recursive2, queries kept)
CHAIN = [
    ("proof_recursive2_01.json", "/finalProof/recursive2.zkin.proof_01.json", True, QUERIES_OUT),
    ("proof_recursive1_0.json", "/aggregatedProof/recursive1.zkin.proof_0.json", False, QUERIES_OUT),
    ("proof_recursive1_1.json", "/aggregatedProof/recursive1.zkin.proof_1.json", False, [0, 42]),
    ("proof_recursive1_2.json", "/aggregatedProof/recursive1.zkin.proof_2.json", False, [0, 42]),
    ("proof_recursive1_3.json", "/aggregatedProof/recursive1.zkin.proof_3.json", False, [0, 42]),
    ("proof_recursive2_23.json", "/finalProof/recursive2.zkin.proof_23.json", True, [0, 42]),
    ("proof_recursive2_03.json", "/finalProof/recursive2.zkin.proof_03.json", True, [0, 42]),
]
# which two proofs the reference joined into which (aggregator_service.cpp:31-144; Prover::genAggregatedProof prover.cpp:674-783)
JOINS = [("proof_recursive1_0.json", "proof_recursive1_1.json", "proof_recursive2_01.json"),
         ("proof_recursive1_2.json", "proof_recursive1_3.json", "proof_recursive2_23.json"),
         ("proof_recursive2_01.json", "proof_recursive2_23.json", "proof_recursive2_03.json")]


def main():
    kats = poseidon_kats()
    json.dump(kats, open(os.path.join(HERE, "poseidon_kats.json"), "w"), separators=(",", ":"))
    const2, const1, pubs = None, None, {}
    for fname, src, is_r2, qs in CHAIN:
        fx, const = golden_proof(REF + src, None if is_r2 else const2, src.rsplit("/", 1)[1][:-5].replace(".zkin", ""), qs)
        if is_r2:      # every recursive2 proof was made with the same constant tree (one circuit), likewise recursive1
            assert const2 is None or const == const2, (const, const2)
            const2 = const
        else:
            assert const1 is None or const == const1, (const, const1)
            const1 = const
        pubs[fname] = fx["publics"][:44]
        json.dump(fx, open(os.path.join(HERE, fname), "w"), separators=(",", ":"))
    # the aggregation chain as the reference produced it: publics of the joined proof = joinzkin's stitch of its two inputs
    # (proof2zkinStark.cpp:84-125) and the inputs pass genAggregatedProof's consistency checks (prover.cpp:695-728)
    for a, b, out in JOINS:
        pa, pb, po = pubs[a], pubs[b], pubs[out]
        assert po == pa[:19] + pb[19:44], (a, b, out)
        assert pa[17] == pb[17] and pa[18] == pb[18] and pa[19:27] == pb[0:8] and pa[27:35] == pb[8:16] and pa[43] == pb[16], (a, b)
    print("aggregation chain: 3 joins reproduce the golden aggregated publics; all inputs pass the reference's consistency checks")
    json.dump({"source": "publics of the seven golden proofs (testvectors/aggregatedProof, testvectors/finalProof) and which pairs the reference "
                         "aggregated into which (test/service/aggregator/aggregator_service.cpp:31-144)",
               "recursive1_constRoot": [str(x) for x in const1], "recursive2_constRoot": [str(x) for x in const2],
               "joins": [{"a": a, "b": b, "out": out} for a, b, out in JOINS]},
              open(os.path.join(HERE, "aggregation_chain.json"), "w"), indent=1)
    # the four G1 commitments of the reference's fflonk proof: real BN254 curve points held by the reference (pin q and b = 3)
    blob = open(REF + "/finalProof/proof.json").read().strip()[2:]
    w = [int(blob[64 * i:64 * i + 64], 16) for i in range(len(blob) // 64)]
    assert len(w) == 24 and len(blob) == 24 * 64
    json.dump({"source": "testvectors/finalProof/proof.json: 24 32-byte words = the G1 commitments C1, C2, W1, W2 of the reference's fflonk proof (affine x, y), "
                         "then its 16 evaluations",
               "points": [[str(w[2 * i]), str(w[2 * i + 1])] for i in range(4)], "evaluations": [str(x) for x in w[8:]]},
              open(os.path.join(HERE, "fflonk_proof_g1_points.json"), "w"), indent=1)
    # BN254 field constants the reference's generated field code holds as DATA (src/ffiasm/fr.asm / fq.asm, section .data): modulus, R^2, R^3 and
    # np = -q^-1 mod 2^64 of the 4 x 64-bit Montgomery form.  They pin the element representation (R = 2^256) of RawFr / RawFq -- what "the bytes of
    # a RawFr::Element" means -- for the oracle (oracle/bn128_oracle.py) and for the limbs / N0INV the kernels are built with.
    import re

    def asm_consts(path):
        text = open(path).read()
        data = text[text.index("section .data"):]
        out = {}
        for name in ("q", "half", "R2", "R3", "np"):
            m = re.search(r"^%s\s+dq\s+([0-9a-fx, ]+)$" % name, data, re.M)
            out[name] = [v.strip() for v in m.group(1).split(",")]
        return out
    json.dump({"source": "src/ffiasm/fr.asm and src/ffiasm/fq.asm, `section .data`: little-endian 64-bit limbs as written there",
               "Fr": asm_consts("/root/reference/src/ffiasm/fr.asm"), "Fq": asm_consts("/root/reference/src/ffiasm/fq.asm")},
              open(os.path.join(HERE, "bn254_field_constants.json"), "w"), indent=1)
    for f in ["poseidon_kats.json", "fflonk_proof_g1_points.json", "aggregation_chain.json", "bn254_field_constants.json"] + [c[0] for c in CHAIN]:
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
