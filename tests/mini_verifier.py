"""Verifier of the toy STARK proved by tests/mini_genproof_flow.py -- plain Python integers, no prover code:
transcript replay, Merkle paths, the constraint identity at xi, the FRI polynomial at every query from the
opened rows, and the fold chain (orc.fri_fold_one and orc.merkle_root_from_path are the two oracle pieces used;
both are pinned to the reference's golden proofs in tests/test_oracle_golden.py).  An output that passes is a
valid proof whatever produced it, so this checks the MEANING of LEv/evmap, the quotient split, xDivXSubXi and
the FRI polynomial program, not only that two implementations agree."""
from tests import mini_genproof_flow as flow

P = flow.P


def add3(a, b): return [(x + y) % P for x, y in zip(a, b)]
def sub3(a, b): return [(x - y) % P for x, y in zip(a, b)]
def emb(x): return [x % P, 0, 0]


def mul3(a, b):   # Fp[x]/(x^3 - x - 1)
    a0, a1, a2 = a; b0, b1, b2 = b
    return [(a0 * b0 + a1 * b2 + a2 * b1) % P,
            (a0 * b1 + a1 * b0 + a1 * b2 + a2 * b1 + a2 * b2) % P,
            (a0 * b2 + a2 * b0 + a1 * b1 + a2 * b2) % P]


def pow3(a, e):
    r = [1, 0, 0]
    while e:
        if e & 1: r = mul3(r, a)
        a = mul3(a, a); e >>= 1
    return r


def inv3(orc, a): return [int(v) for v in orc.inv3(a)]


def verify(orc, proof, nbits, fri_steps, nqueries, gl_w):
    N, NE = 1 << nbits, 1 << (nbits + 1)
    L = flow.Layout(nbits)
    _, evmap = flow.programs(L)
    shift, w_n, w_ne = 7, gl_w(nbits), gl_w(nbits + 1)
    # ---- transcript
    t = flow.HostTranscript(lambda st: orc.perm(st))
    t.put(proof["verkey"]); t.put(proof["publics"])
    t.put(proof["root1"]); u = t.get_field(); t.get_field()
    t.put(proof["root2"]); gamma = t.get_field(); beta = t.get_field()
    t.put(proof["root3"]); vc = t.get_field()
    t.put(proof["root4"]); xi = t.get_field()
    for e in proof["evals"]:
        t.put(e)
    v1 = t.get_field(); v2 = t.get_field()
    fri = proof["fri"]
    special = []
    for si in range(len(fri_steps)):
        special.append(t.get_field())
        if si < len(fri_steps) - 1:
            t.put(fri["roots"][si])
        else:
            t.put(fri["finalPol"])
    ys = t.get_permutations(nqueries, fri_steps[0])
    assert ys == fri["queries"], "query indices"
    # ---- constraint identity at xi
    ev = {(s, c, pr): e for (s, c, d, pr), e in zip(evmap, proof["evals"])}
    a, b = ev[("cm1_2ns", 0, False)], ev[("cm1_2ns", 1, False)]
    h1, h2, z = ev[("cm2_2ns", 0, False)], ev[("cm2_2ns", 3, False)], ev[("cm3_2ns", 0, False)]
    q0, q1 = ev[("cm4_2ns", 0, False)], ev[("cm4_2ns", 3, False)]
    T0, T1, L1 = ev[("const", 0, False)], ev[("const", 1, False)], ev[("const", 2, False)]
    h1p, zp, T0p, T1p = ev[("cm2_2ns", 0, True)], ev[("cm3_2ns", 0, True)], ev[("const", 0, True)], ev[("const", 1, True)]
    one = [1, 0, 0]
    f = add3(a, mul3(u, b)); tt = add3(T0, mul3(u, T1)); ttp = add3(T0p, mul3(u, T1p))
    ob = add3(one, beta); g1 = mul3(gamma, ob)
    num = mul3(mul3(add3(gamma, f), ob), add3(add3(mul3(ttp, beta), tt), g1))
    den = mul3(add3(add3(mul3(h2, beta), h1), g1), add3(add3(mul3(h1p, beta), h2), g1))
    C = add3(sub3(b, mul3(a, a)), mul3(vc, sub3(mul3(zp, den), mul3(z, num))))
    C = add3(C, mul3(mul3(vc, vc), mul3(L1, sub3(z, one))))
    xiN = pow3(xi, N)
    Q = add3(q0, mul3(xiN, q1))
    assert mul3(Q, sub3(xiN, one)) == C, "constraint identity C(xi) = Q(xi) Zh(xi)"
    # ---- queries
    roots0 = [proof["root1"], proof["root2"], proof["root3"], proof["root4"], proof["verkey"]]
    plain = [e for e in evmap if not e[3]]; prime = [e for e in evmap if e[3]]
    tree_of = {"cm1_2ns": 0, "cm2_2ns": 1, "cm3_2ns": 2, "cm4_2ns": 3, "const": 4}
    wxi = mul3(xi, emb(w_n))
    sinvs = [None, pow(shift, P - 2, P)]
    for si in range(1, len(fri_steps)):
        s = sinvs[si]
        for _ in range(fri_steps[si - 1] - fri_steps[si]):
            s = s * s % P
        sinvs.append(s)
    for qi, y in enumerate(ys):
        rows = []
        for k, (row, sib) in enumerate(fri["s0"][qi]):
            r = orc.merkle_root_from_path(row, sib, y)
            assert [int(v) for v in r] == roots0[k], ("step-0 path", k)
            rows.append(row)
        x = shift * pow(w_ne, y, P) % P

        def val(e):
            s, c, d, _ = e
            r = rows[tree_of[s]]
            return [r[c], r[c + 1], r[c + 2]] if d == 3 else emb(r[c])

        def horner(lst, base):
            acc = None
            for k, e in enumerate(lst):
                term = sub3(val(e), proof["evals"][base + k])
                acc = term if acc is None else add3(mul3(acc, v1), term)
            return acc
        A, B = horner(plain, 0), horner(prime, len(plain))
        xd = mul3(inv3(orc, sub3(emb(x), xi)), emb(x)); xdw = mul3(inv3(orc, sub3(emb(x), wxi)), emb(x))
        ffri = add3(mul3(A, xd), mul3(v2, mul3(B, xdw)))
        # fold chain: tree si row g holds the 2^(steps[si-1]-steps[si]) values that fold to index g of layer si
        idx = y
        expect = ffri                                    # layer-0 value at y (step 0 of FRIProve::prove is a copy)
        for si in range(1, len(fri_steps)):
            ng = 1 << fri_steps[si]
            g = idx % ng
            row, sib = fri["si"][si - 1][qi]
            r = orc.merkle_root_from_path(row, sib, g)
            assert [int(v) for v in r] == fri["roots"][si - 1], ("fri path", si)
            k = idx >> fri_steps[si]
            assert row[3 * k: 3 * k + 3] == expect, ("layer value", si, qi)
            bits = fri_steps[si - 1]
            expect = [int(v) for v in orc.fri_fold_one(row, bits, bits - fri_steps[si], g, special[si], sinvs[si])]
            idx = g
        assert fri["finalPol"][3 * idx: 3 * idx + 3] == expect, ("final polynomial", qi)
    # ---- the final polynomial has degree < 2^(last - 1) (blow-up 2): the upper half of its coefficients is zero
    import numpy as np
    last = fri_steps[-1]
    co = orc.ntt(np.array(fri["finalPol"], np.uint64), 1 << last, 3, inverse=True)
    assert not co[1 << (last - 1):].any() and co[: 1 << (last - 1)].any(), "final polynomial degree"
    return True
