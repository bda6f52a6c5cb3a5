"""A complete (toy) STARK prover in the stage order of Starks::genProof (src/starkpil/starks.cpp:9-404), written
ONCE against a small backend interface and run on two backends: the HIP library (through the C ABI) and the CPU
oracle.  Every stage the reference runs is present -- witness commit, plookup H1/H2, grand-product Z, quotient
(expression program on the extended domain, INTT / coefficient split / NTT), evaluations (LEv, LpEv, evmap),
xDivXSubXi, FRI polynomial (expression program), FRI -- so the two runs must agree on every root, evaluation, FRI
layer and opened row.  Tests / benchmarks only (it drives the oracle as one of its backends).

The proved system ("mini PIL", N = 2^nBits rows, blow-up 2):
    cm1: a, b            const: T0[i] = i, T1[i] = i^2, L1 = (1, 0, 0, ...)
    lookup  (a, b) in (T0, T1) compressed with u: f = a + u b, t = T0 + u T1        -> h1, h2 (cm2, ext-3)
    z (cm3, ext-3): z[0] = 1, z' den = z num with
        num = (1 + beta)(gamma + f)(gamma (1 + beta) + t + beta t'),  den = (gamma(1+beta) + h1 + beta h2)(gamma(1+beta) + h2 + beta h1')
    q = (b - a a  +  vc (z' den - z num)  +  vc^2 L1 (z - 1)) / Zh                     -> cm4 (qDeg 2 x ext-3)
    f_fri = A xDivXSubXi + v2 B xDivXSubWXi,  A / B = Horner_v1(pol - eval) over the plain / next-row openings
The slab layout follows stark_info.cpp (mapOffsets / mapSectionsN); challenges are numbered as in starks.cpp
(0 u, 1 defVal, 2 gamma, 3 beta, 4 vc, 5 v1, 6 v2, 7 xi)."""
import numpy as np

from xlayer_prover_amd import expr as E
from xlayer_prover_amd.transcript import HostTranscript  # noqa: F401  (used by the flow and re-exported)

P = 0xFFFFFFFF00000001
W1, W2, W3, WT, W4, WC = 2, 6, 3, 12, 6, 3
QDEG, QDIM = 2, 3


class Layout:
    """slab offsets in elements (cm*_n, tmpExp_n, cm*_2ns, q_2ns, f_2ns), as mapOffsets would give them"""

    def __init__(self, nbits):
        self.nbits, self.nbits_ext = nbits, nbits + 1
        self.N, self.NE = 1 << nbits, 1 << (nbits + 1)
        o = 0
        self.off = {}
        for name, w, rows in (("cm1_n", W1, self.N), ("cm2_n", W2, self.N), ("cm3_n", W3, self.N), ("tmpExp_n", WT, self.N),
                              ("cm1_2ns", W1, self.NE), ("cm2_2ns", W2, self.NE), ("cm3_2ns", W3, self.NE), ("cm4_2ns", W4, self.NE)):
            self.off[name] = (o, w, rows)
            o += w * rows
        self.total = o


def programs(L):
    """the five step programs of the mini PIL (xlp expression VM encoding)"""
    N, NE = L.N, L.NE
    o = L.off
    ch = E.challenge

    def col(sec, c, dim=1): return E.pols(o[sec][0] + c, o[sec][1], dim)
    def colp(sec, c, nxt, n, dim=1): return E.pols_prime(o[sec][0] + c, nxt, n, o[sec][1], dim)
    t3, t1 = E.tmp3, E.tmp1
    # tmpExp_n columns: f 0..2, t 3..5, num 6..8, den 9..11
    s2 = E.Program()
    s2.mul(t3(0), col("cm1_n", 1), ch(0)).add(col("tmpExp_n", 0, 3), col("cm1_n", 0), t3(0))          # f = a + u b
    s2.mul(t3(0), E.const(1), ch(0)).add(col("tmpExp_n", 3, 3), E.const(0), t3(0))                     # t = T0 + u T1

    def num_den(p, f, t, tp, h1, h2, h1p, out_num, out_den):
        # g1 = gamma (1 + beta)
        p.add(t3(1), E.number(1), ch(3)).mul(t3(2), t3(1), ch(2))                                       # t3(1) = 1+beta, t3(2) = g1
        p.add(t3(3), f, ch(2)).mul(t3(3), t3(3), t3(1))                                                 # (gamma + f)(1 + beta)
        p.mul(t3(4), tp, ch(3)).add(t3(4), t3(4), t).add(t3(4), t3(4), t3(2))                           # g1 + t + beta t'
        p.mul(out_num, t3(3), t3(4))
        p.mul(t3(3), h2, ch(3)).add(t3(3), t3(3), h1).add(t3(3), t3(3), t3(2))                          # g1 + h1 + beta h2
        p.mul(t3(4), h1p, ch(3)).add(t3(4), t3(4), h2).add(t3(4), t3(4), t3(2))                         # g1 + h2 + beta h1'
        p.mul(out_den, t3(3), t3(4))

    s3p = E.Program()
    num_den(s3p, col("tmpExp_n", 0, 3), col("tmpExp_n", 3, 3), colp("tmpExp_n", 3, 1, N, 3), col("cm2_n", 0, 3), col("cm2_n", 3, 3),
            colp("cm2_n", 0, 1, N, 3), col("tmpExp_n", 6, 3), col("tmpExp_n", 9, 3))
    s3 = E.Program()                                                                                     # nothing to add in this PIL
    s3.copy(t1(0), E.number(0))
    # ---- extended domain: rebuild f, t, t', num, den from the extended sections, then the quotient
    s4 = E.Program()
    s4.mul(t3(0), col("cm1_2ns", 1), ch(0)).add(t3(5), col("cm1_2ns", 0), t3(0))                        # f
    s4.mul(t3(0), E.const2ns(1), ch(0)).add(t3(6), E.const2ns(0), t3(0))                                # t
    s4.mul(t3(0), E.const2ns_prime(1, 2, NE), ch(0)).add(t3(7), E.const2ns_prime(0, 2, NE), t3(0))      # t'
    num_den(s4, t3(5), t3(6), t3(7), col("cm2_2ns", 0, 3), col("cm2_2ns", 3, 3), colp("cm2_2ns", 0, 2, NE, 3), t3(8), t3(9))
    s4.mul(t3(8), t3(8), col("cm3_2ns", 0, 3)).mul(t3(9), t3(9), colp("cm3_2ns", 0, 2, NE, 3)).sub(t3(8), t3(9), t3(8))   # z' den - z num
    s4.mul(t3(8), t3(8), ch(4))
    s4.mul(t1(0), col("cm1_2ns", 0), col("cm1_2ns", 0)).sub(t1(0), col("cm1_2ns", 1), t1(0)).add(t3(8), t1(0), t3(8))      # + (b - a^2)
    s4.sub(t3(9), col("cm3_2ns", 0, 3), E.number(1)).mul(t3(9), t3(9), E.const2ns(2)).mul(t3(9), t3(9), ch(4)).mul(t3(9), t3(9), ch(4))
    s4.add(t3(8), t3(8), t3(9)).mul(E.q_2ns(), E.zi(), t3(8))
    # ---- FRI polynomial
    plain = [("cm1_2ns", 0, 1), ("cm1_2ns", 1, 1), ("cm2_2ns", 0, 3), ("cm2_2ns", 3, 3), ("cm3_2ns", 0, 3), ("cm4_2ns", 0, 3), ("cm4_2ns", 3, 3),
             ("const", 0, 1), ("const", 1, 1), ("const", 2, 1)]
    prime = [("cm2_2ns", 0, 3), ("cm3_2ns", 0, 3), ("const", 0, 1), ("const", 1, 1)]
    evmap = [(s, c, d, False) for s, c, d in plain] + [(s, c, d, True) for s, c, d in prime]
    s5 = E.Program()

    def horner(lst, base, acc):
        first = True
        for k, (sec, c, d) in enumerate(lst):
            src = E.const2ns(c) if sec == "const" else col(sec, c, d)
            if not first:
                s5.mul(acc, acc, ch(5))
            s5.sub(t3(1), src, E.evaluation(base + k))
            if first:
                s5.copy(acc, t3(1)); first = False
            else:
                s5.add(acc, acc, t3(1))
    horner(plain, 0, t3(0))
    s5.mul(t3(0), t3(0), E.x_div_x_sub_xi())
    horner(prime, len(plain), t3(2))
    s5.mul(t3(2), t3(2), E.x_div_x_sub_wxi()).mul(t3(2), t3(2), ch(6)).add(E.f_2ns(), t3(0), t3(2))
    return {"step2prev": s2, "step3prev": s3p, "step3": s3, "step42ns": s4, "step52ns": s5}, evmap


def toy_stark_files(outdir, name, nbits, fri_steps, nqueries, seed, optimise=False):
    """Everything the C++ `Starks` mirror (xlayer_prover_amd/host/starks.hpp) needs to prove the mini PIL: a starkinfo.json
    in the reference's schema (stark_info.cpp:21-520), the five step programs, the constant and committed polynomials
    and the publics -- so the C++ genProof can be compared with this module's flow."""
    import json
    import os
    import struct
    L = Layout(nbits)
    N, NE = L.N, L.NE
    progs, evmap = programs(L)
    off = {k: v[0] for k, v in L.off.items()}
    end = L.total
    off["cm4_n"] = 0
    off["q_2ns"] = end; end += 3 * NE
    off["f_2ns"] = end; end += 3 * NE
    widths = {"cm1_n": W1, "cm2_n": W2, "cm3_n": W3, "cm4_n": 0, "tmpExp_n": WT, "cm1_2ns": W1, "cm2_2ns": W2, "cm3_2ns": W3, "cm4_2ns": W4, "q_2ns": 3, "f_2ns": 3}
    secs = list(widths)
    var = [("cm1_n", 1, 0), ("cm1_n", 1, 1), ("cm2_n", 3, 0), ("cm2_n", 3, 3), ("cm3_n", 3, 0),
           ("tmpExp_n", 3, 0), ("tmpExp_n", 3, 3), ("tmpExp_n", 3, 6), ("tmpExp_n", 3, 9),
           ("cm1_2ns", 1, 0), ("cm1_2ns", 1, 1), ("cm2_2ns", 3, 0), ("cm2_2ns", 3, 3), ("cm3_2ns", 3, 0), ("cm4_2ns", 3, 0), ("cm4_2ns", 3, 3)]
    cmid = {("cm1_2ns", 0): 0, ("cm1_2ns", 1): 1, ("cm2_2ns", 0): 2, ("cm2_2ns", 3): 3, ("cm3_2ns", 0): 4}
    qid = {("cm4_2ns", 0): 0, ("cm4_2ns", 3): 1}
    ev = []
    for s_, c, d, pr in evmap:
        if s_ == "const": ev.append({"type": "const", "id": c, "prime": pr})
        elif (s_, c) in cmid: ev.append({"type": "cm", "id": cmid[(s_, c)], "prime": pr})
        else: ev.append({"type": "q", "id": qid[(s_, c)], "prime": pr})
    step = {"tmpUsed": 0, "first": [], "i": [], "last": []}
    j = {"starkStruct": {"nBits": nbits, "nBitsExt": nbits + 1, "nQueries": nqueries, "verificationHashType": "GL", "steps": [{"nBits": b} for b in fri_steps]},
         "mapTotalN": end, "nConstants": WC, "nPublics": 4, "nCm1": 2, "nCm2": 2, "nCm3": 1, "nCm4": 2, "friExpId": 0, "nExps": 0, "qDim": QDIM, "qDeg": QDEG,
         "mapDeg": {s_: (N if s_.endswith("_n") else NE) for s_ in secs}, "mapOffsets": {s_: off[s_] for s_ in secs},
         "mapSections": {s_: [] for s_ in secs}, "mapSectionsN": widths, "mapSectionsN1": widths, "mapSectionsN3": {s_: 0 for s_ in secs},
         "varPolMap": [{"section": a, "dim": d, "sectionPos": p_} for a, d, p_ in var], "qs": [14, 15], "cm_n": [0, 1, 2, 3, 4], "cm_2ns": [9, 10, 11, 12, 13],
         "peCtx": [], "ciCtx": [],
         "puCtx": [{"tExpId": 101, "fExpId": 100, "h1Id": 2, "h2Id": 3, "zId": 4, "c1Id": 0, "numId": 102, "denId": 103, "c2Id": 0}],
         "evMap": ev, "step2prev": step, "step3prev": step, "step3": step, "step42ns": step, "step52ns": step,
         "exps_n": [], "q_2ns": [], "cm4_n": [], "cm4_2ns": [], "tmpExp_n": [], "exp2pol": {"100": 5, "101": 6, "102": 7, "103": 8}}
    os.makedirs(outdir, exist_ok=True)
    json.dump(j, open(os.path.join(outdir, name + ".starkinfo.json"), "w"))
    magic = int.from_bytes(b"XLPEXPR1", "little")
    for stepname, prog in progs.items():
        with open(os.path.join(outdir, f"{name}.{stepname}.xlpexpr"), "wb") as f:
            code = prog.code
            f.write(struct.pack("<4Q", magic, prog.ntmp1, prog.ntmp3, code.size)); f.write(code.astype("<u8").tobytes())
    if optimise:                                          # dot-product forms next to the programs (tools/optimize_programs.py)
        from tools.optimize_programs import optimise_file
        wrote = [optimise_file(os.path.join(outdir, f"{name}.{stepname}.xlpexpr"))[1] for stepname in progs]
        assert any(wrote)
    cm1, const, publics = witness(L, seed)
    cm1.astype("<u8").tofile(os.path.join(outdir, name + ".commit"))
    const.astype("<u8").tofile(os.path.join(outdir, name + ".const"))
    publics.astype("<u8").tofile(os.path.join(outdir, name + ".publics"))


def witness(L, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, L.N, L.N).astype(np.uint64)
    b = (a * a) % np.uint64(P)
    cm1 = np.stack([a, b], axis=1).reshape(-1)
    i = np.arange(L.N, dtype=np.uint64)
    l1 = np.zeros(L.N, np.uint64); l1[0] = 1
    const = np.stack([i, (i * i) % np.uint64(P), l1], axis=1).reshape(-1)
    publics = rng.integers(0, 1 << 62, 4).astype(np.uint64)
    return cm1, const, publics


def _mul(a, b): return a * b % P
def _inv(a): return pow(a, P - 2, P)


def _mul3s(a, s): return [x * s % P for x in a]


def fri_prove(B, tr, f_2ns, fri_steps, nqueries, shift, open_step0):
    """FRIProve::prove (friProve.cpp:5-190) on the full FRI polynomial; open_step0(ys) returns, per query, the opened rows +
    sibling paths of the five step-0 trees (friProve.cpp:150-177)."""
    res = {"roots": [], "queries": None}
    pol = f_2ns; bits = fri_steps[0]; sinv = _inv(shift)
    trees = []
    for si, nb in enumerate(fri_steps):
        red = bits - nb
        x = tr.get_field()
        pol2 = B.fri_fold(pol, bits, red, x, sinv)
        if si < len(fri_steps) - 1:
            ng = 1 << fri_steps[si + 1]; gs = (1 << nb) // ng
            aux = B.transpose3(pol2, 1 << nb, fri_steps[si + 1])
            nodes = B.merkle(aux, gs * 3, ng)
            trees.append((aux, nodes, gs * 3, ng))
            r = B.root(nodes, ng); res["roots"].append(r); tr.put(r)
        else:
            fin = B.to_host(pol2)[: 3 << nb]
            res["finalPol"] = [int(v) for v in fin]; tr.put(res["finalPol"])
        pol = pol2; bits = nb
        for _ in range(red):
            sinv = sinv * sinv % P
    ys = tr.get_permutations(nqueries, fri_steps[0])
    res["queries"] = ys
    res["s0"] = open_step0(ys)
    res["si"] = []
    idx = ys
    for si, (aux, nodes, w, ng) in enumerate(trees):
        idx = [y % ng for y in idx]
        res["si"].append([B.proof(aux, nodes, w, ng, y) for y in idx])
    return res


def prove(B, nbits, fri_steps, nqueries, seed=1, timers=None, debug=False):
    """B: backend (see tests/test_mini_genproof.py).  Returns a dict of everything a verifier would read."""
    import time
    L = Layout(nbits)
    N, NE = L.N, L.NE
    progs, evmap = programs(L)
    cm1_h, const_h, publics = witness(L, seed)
    w_n, w_ne, shift = B.w(nbits), B.w(nbits + 1), 7

    def timed(name, fn):
        if timers is None:
            return fn()
        B.sync(); t0 = time.perf_counter(); r = fn(); B.sync()
        timers[name] = timers.get(name, 0.0) + time.perf_counter() - t0
        return r

    # ---- setup (Starks ctor, starks.hpp:81-190): slab, const pols + their extension and tree, x tables, ZhInv
    pols = B.zeros(L.total)
    def sec(name):
        o, w, rows = L.off[name]
        return pols[o:o + w * rows]
    B.copy_into(sec("cm1_n"), B.from_host(cm1_h))
    const_n = B.from_host(const_h)
    const_2ns = B.zeros(NE * WC)
    B.extend(const_2ns, const_n, NE, N, WC)
    nodesC = B.merkle(const_2ns, WC, NE)
    x_n = B.geometric(N, 1, [1], [w_n])
    x_2ns = B.geometric(NE, 1, [shift], [w_ne])
    sn = pow(shift, N, P)                                                   # ZhInv (zhInv.cpp:7-32)
    zi = B.from_host(np.array([_inv((sn * pow(B.w(1), i, P) - 1) % P) for i in range(2)], np.uint64))
    q_2ns = B.zeros(NE * 3); f_2ns = B.zeros(NE * 3)
    challenges = np.zeros(8 * 3, np.uint64)
    evals = np.zeros(len(evmap) * 3, np.uint64)
    xdx = B.zeros(NE * 3); xdwx = B.zeros(NE * 3)

    def run(step, nrows):
        bufs = {"pols": pols, "const_pols": const_n, "const_pols_2ns": const_2ns, "challenges": B.from_host(challenges),
                "public_inputs": B.from_host(publics), "evals": B.from_host(evals), "x_n": x_n, "x_2ns": x_2ns, "zi": zi,
                "x_div_x_sub_xi": xdx, "x_div_x_sub_wxi": xdwx, "q_2ns": q_2ns, "f_2ns": f_2ns}
        B.expr_run(progs[step], bufs, WC, WC, 0, nrows)

    out = {"publics": [int(v) for v in publics]}
    tr = B.transcript() if hasattr(B, "transcript") else HostTranscript(B.perm)     # StarkRecursiveF: TranscriptBN128
    verkey = B.root(nodesC, NE)
    tr.put(verkey); tr.put(publics)
    # ---- 1
    timed("STEP_1_LDE", lambda: B.extend(sec("cm1_2ns"), sec("cm1_n"), NE, N, W1))
    nodes1 = timed("STEP_1_MERKLETREE", lambda: B.merkle(sec("cm1_2ns"), W1, NE))
    out["root1"] = B.root(nodes1, NE); tr.put(out["root1"])
    # ---- 2: plookup h1, h2
    challenges[0:3] = tr.get_field(); challenges[3:6] = tr.get_field()
    timed("STEP_2_CALCULATE_EXPS", lambda: run("step2prev", N))
    def h1h2():
        f = B.get_cols(sec("tmpExp_n"), N, WT, 0, 3); t = B.get_cols(sec("tmpExp_n"), N, WT, 3, 3)
        h1 = B.zeros(N * 3); h2 = B.zeros(N * 3)
        B.h1h2(h1, h2, f, t, N, 3)
        B.set_cols(sec("cm2_n"), N, W2, 0, 3, h1); B.set_cols(sec("cm2_n"), N, W2, 3, 3, h2)
    timed("STEP_2_CALCULATEH1H2", h1h2)
    timed("STEP_2_LDE", lambda: B.extend(sec("cm2_2ns"), sec("cm2_n"), NE, N, W2))
    nodes2 = timed("STEP_2_MERKLETREE", lambda: B.merkle(sec("cm2_2ns"), W2, NE))
    out["root2"] = B.root(nodes2, NE); tr.put(out["root2"])
    # ---- 3: grand product
    challenges[6:9] = tr.get_field(); challenges[9:12] = tr.get_field()
    timed("STEP_3_CALCULATE_EXPS", lambda: run("step3prev", N))
    def calc_z():
        num = B.get_cols(sec("tmpExp_n"), N, WT, 6, 3); den = B.get_cols(sec("tmpExp_n"), N, WT, 9, 3)
        z = B.zeros(N * 3)
        B.calc_z(z, num, den, N)
        B.set_cols(sec("cm3_n"), N, W3, 0, 3, z)
    timed("STEP_3_CALCULATE_Z", calc_z)
    timed("STEP_3_CALCULATE_EXPS_2", lambda: run("step3", N))
    timed("STEP_3_LDE", lambda: B.extend(sec("cm3_2ns"), sec("cm3_n"), NE, N, W3))
    nodes3 = timed("STEP_3_MERKLETREE", lambda: B.merkle(sec("cm3_2ns"), W3, NE))
    out["root3"] = B.root(nodes3, NE); tr.put(out["root3"])
    # ---- 4: quotient
    challenges[12:15] = tr.get_field()
    timed("STEP_4_CALCULATE_EXPS_2NS", lambda: run("step42ns", NE))
    def q_poly():
        qq1 = B.zeros(NE * QDIM)
        B.ntt(qq1, q_2ns, NE, QDIM, True)
        qq2 = B.q_split(qq1, N, NE, QDEG, QDIM, pow(_inv(shift), N, P))
        B.ntt(sec("cm4_2ns"), qq2, NE, QDIM * QDEG, False)
    timed("STEP_4_CALCULATE_EXPS_2NS_INTT_MUL_NTT", q_poly)
    nodes4 = timed("STEP_4_MERKLETREE", lambda: B.merkle(sec("cm4_2ns"), W4, NE))
    out["root4"] = B.root(nodes4, NE); tr.put(out["root4"])
    # ---- 5: evaluations and the FRI polynomial
    xi = tr.get_field(); challenges[21:24] = xi
    def lev():
        xis = _mul3s(xi, _inv(shift)); wxis = _mul3s(_mul3s(xi, w_n), _inv(shift))
        lev_ = B.geometric(N, 3, [1, 0, 0], xis); lpev = B.geometric(N, 3, [1, 0, 0], wxis)
        B.ntt(lev_, lev_, N, 3, True); B.ntt(lpev, lpev, N, 3, True)
        return lev_, lpev
    LEv, LpEv = timed("STEP_5_LEv_LpEv", lev)
    def do_evmap():
        ents = []
        for s, c, d, pr in evmap:
            if s == "const":
                ents.append((const_2ns, WC, c, d, pr))
            else:
                ents.append((sec(s), L.off[s][1], c, d, pr))
        return B.evmap(LEv, LpEv, N, 1, ents)
    ev = timed("STEP_5_EVMAP", do_evmap)
    evals[:] = np.asarray(ev, np.uint64).reshape(-1)
    out["evals"] = [[int(v) for v in row] for row in np.asarray(ev).reshape(-1, 3)]
    for row in out["evals"]:
        tr.put(row)
    challenges[15:18] = tr.get_field(); challenges[18:21] = tr.get_field()
    def xdiv():
        B.copy_into(xdx, B.x_div(x_2ns, xi, NE)); B.copy_into(xdwx, B.x_div(x_2ns, _mul3s(xi, w_n), NE))
    timed("STEP_5_XDIVXSUB", xdiv)
    timed("STEP_5_CALCULATE_EXPS", lambda: run("step52ns", NE))
    # ---- FRI (friProve.cpp:5-190)
    step0 = [(sec("cm1_2ns"), nodes1, W1), (sec("cm2_2ns"), nodes2, W2), (sec("cm3_2ns"), nodes3, W3), (sec("cm4_2ns"), nodes4, W4), (const_2ns, nodesC, WC)]
    open_step0 = lambda ys: [[B.proof(src, nd, w, NE, y) for (src, nd, w) in step0] for y in ys]
    fri = lambda: fri_prove(B, tr, f_2ns, fri_steps, nqueries, shift, open_step0)
    out["fri"] = timed("STEP_FRI", fri)
    out["verkey"] = verkey
    if debug:
        out["_state"] = (B.to_host(pols), L)
    return out
