"""Oracle-side rebuild of xlayer_prover_amd/host/starks_mini.hpp (test infrastructure):
the same synthetic mini-STARK (commit stages + FRIProve::prove in the order of
starks.cpp / friProve.cpp) computed with the CPU oracle only."""
import numpy as np


def mini_stark_oracle(orc, nBits, nBitsExt, nQueries, steps, w1=18, w3=39, w4=21, wC=52, nEvals=10, nPublics=8):
    N, NE = 1 << nBits, 1 << nBitsExt
    fill = lambda n, seed: orc.fill(n, seed)
    cm1_n = fill(N * w1, 0x5EED0101).reshape(N, w1)
    cm3_n = fill(N * w3, 0x5EED0103).reshape(N, w3)
    const_n = fill(N * wC, 0x5EED010C).reshape(N, wC)
    qq2 = fill(NE * w4, 0x5EED0104).reshape(NE, w4)
    f_2ns = fill(NE * 3, 0x5EED010F).reshape(NE, 3)
    evals = fill(nEvals * 3, 0x5EED010E).reshape(nEvals, 3)
    publics = fill(nPublics, 0x5EED0100)

    srcs = [orc.extend_pol(cm1_n, NE, N, w1), np.zeros((NE, 0), np.uint64), orc.extend_pol(cm3_n, NE, N, w3),
            orc.ntt(qq2, NE, w4), orc.extend_pol(const_n, NE, N, wC)]
    widths = [w1, 0, w3, w4, wC]
    trees = [orc.merkletree(s, w, NE) for s, w in zip(srcs, widths)]
    roots = [t[-4:] for t in trees]
    verkey = roots[4]
    t = orc.Transcript()
    t.put(verkey); t.put(publics)
    t.put(roots[0]); t.get_field(); t.get_field()
    t.put(roots[1]); t.get_field(); t.get_field()
    t.put(roots[2]); t.get_field()
    t.put(roots[3]); t.get_field()
    for e in evals:
        t.put(e)
    t.get_field(); t.get_field()
    # ---- FRI (friProve.cpp:5-190)
    pol, polBits = f_2ns, nBitsExt
    sinv = orc.inv(7)
    fri_trees = [None] * len(steps)   # (src, nodes, width, height)
    out = {}
    for si, nb in enumerate(steps):
        red = polBits - nb
        x = t.get_field()
        pol2 = orc.fri_fold(pol, polBits, red, x, sinv)
        if si < len(steps) - 1:
            nGroups = 1 << steps[si + 1]
            groupSize = (1 << nb) // nGroups
            aux = orc.transpose3(pol2, 1 << nb, steps[si + 1])
            nodes = orc.merkletree(aux, groupSize * 3, nGroups)
            fri_trees[si + 1] = (aux.reshape(-1), nodes, groupSize * 3, nGroups)
            t.put(nodes[-4:])
            out[f"s{si+1}_root"] = [str(int(v)) for v in nodes[-4:]]
        else:
            for e in pol2:
                t.put(e)
        pol, polBits = pol2, nb
        for _ in range(red):
            sinv = orc.mul(sinv, sinv)
    ys = [int(y) for y in t.get_permutations(nQueries, steps[0])]
    S = lambda a: [str(int(v)) for v in a]
    names = ["1", "2", "3", "4", "C"]
    for ti, nm in enumerate(names):
        if widths[ti] == 0:
            continue
        vals, sibs = [], []
        for y in ys:
            pr = orc.merkle_proof(trees[ti], srcs[ti].reshape(-1), widths[ti], NE, y)
            vals.append(S(pr[:widths[ti]]))
            sibs.append([S(pr[widths[ti] + 4 * k: widths[ti] + 4 * k + 4]) for k in range(nBitsExt)])
        out[f"s0_vals{nm}"], out[f"s0_siblings{nm}"] = vals, sibs
    cur = ys
    for si in range(1, len(steps)):
        cur = [y % (1 << steps[si]) for y in cur]
        src, nodes, w, h = fri_trees[si]
        vals, sibs = [], []
        for y in cur:
            pr = orc.merkle_proof(nodes, src, w, h, y)
            vals.append(S(pr[:w]))
            sibs.append([S(pr[w + 4 * k: w + 4 * k + 4]) for k in range(steps[si])])
        out[f"s{si}_vals"], out[f"s{si}_siblings"] = vals, sibs
    out.update({"root1": S(roots[0]), "root2": S(roots[1]), "root3": S(roots[2]), "root4": S(roots[3]),
                "evals": [S(e) for e in evals], "finalPol": [S(e) for e in pol], "publics": S(publics), "verkey": S(verkey)})
    return out
