"""The fork-9 zkEVM slab geometry (section offsets recovered from the arguments of the reference's generated
step tables, src/starkpil/zkevm/chelpers/zkevm.chelpers.step*.parser.hpp) and a relocation of the converted programs to the
same layout with fewer rows -- so that all five steps address ONE consistent slab that fits a single GPU.
Benchmarks / tests only."""
import numpy as np

from xlayer_prover_amd import programs as R
from xlayer_prover_amd import expr as E

LOG_N_REAL = 23
# (section, width, on the extended domain?) in slab order (stark_info.cpp mapOffsets order)
SECTIONS = [("cm1_n", 751, 0), ("cm2_n", 168, 0), ("cm3_n", 408, 0), ("cm4_n", 6, 0), ("tmpExp_n", 389, 0),
            ("cm1_2ns", 751, 1), ("cm2_2ns", 168, 1), ("cm3_2ns", 408, 1), ("cm4_2ns", 6, 1), ("q_2ns", 3, 1), ("f_2ns", 3, 1)]
# The three STARKs of Prover::genBatchProof (prover.cpp:577,610,644).  The other two are small and keep their real geometry
# (the widths are the strides of their converted step programs, the tmpExp width follows from the offset of
# cm1_2ns).  steps / queries of c12a and lookups / zs of all three are what the absent starkinfo would say: defaults only.
CIRCUITS = {
    "zkevm": {"log_n": 23, "ext": 1, "widths": (751, 168, 408, 6, 389), "steps": ("step2prev", "step3prev", "step3", "step42ns", "step52ns"),
              "lookups": 28, "zs": 40, "queries": 128, "fri": None},
    "c12a": {"log_n": 20, "ext": 2, "widths": (18, 0, 78, 12, 6), "steps": ("step3prev", "step3", "step42ns", "step52ns"),
             "lookups": 0, "zs": 2, "queries": 64, "fri": [22, 18, 14, 10, 6]},
    "recursive1": {"log_n": 17, "ext": 3, "widths": (18, 0, 39, 21, 6), "steps": ("step3prev", "step3", "step42ns", "step52ns"),
                   "lookups": 0, "zs": 2, "queries": 43, "fri": [20, 16, 12, 9, 6]},      # FRI steps / queries of the golden recursive proofs
    # the STARK of Prover::genAggregatedProof (prover.cpp:782): same geometry as recursive1, its own constraint programs
    "recursive2": {"log_n": 17, "ext": 3, "widths": (18, 0, 39, 21, 6), "steps": ("step3prev", "step3", "step42ns", "step52ns"),
                   "lookups": 0, "zs": 2, "queries": 43, "fri": [20, 16, 12, 9, 6]},
    # the STARK of Prover::genFinalProof (StarkRecursiveF, prover.cpp:871): Poseidon-BN128 trees / transcript.  N = 2^17, blow-up 16 and the
    # widths are read off its converted programs (cm1_2ns at word 60 * 2^17, next-row distance 16); FRI steps and query count are a choice
    # (its starkstruct is absent from the snapshot)
    "recursivef": {"log_n": 17, "ext": 4, "widths": (12, 0, 21, 21, 6), "steps": ("step3prev", "step3", "step42ns", "step52ns"),
                   "lookups": 0, "zs": 2, "queries": 32, "fri": [21, 17, 13, 9, 5], "hash": "BN128"},
}


def sections(circ="zkevm"):
    w1, w2, w3, w4, wt = CIRCUITS[circ]["widths"]
    return [("cm1_n", w1, 0), ("cm2_n", w2, 0), ("cm3_n", w3, 0), ("cm4_n", w4, 0), ("tmpExp_n", wt, 0),
            ("cm1_2ns", w1, 1), ("cm2_2ns", w2, 1), ("cm3_2ns", w3, 1), ("cm4_2ns", w4, 1), ("q_2ns", 3, 1), ("f_2ns", 3, 1)]
REAL_OFFSETS = {"cm1_n": 0, "cm2_n": 6299844608, "cm3_n": 7709130752, "cm4_n": 11131682816, "tmpExp_n": 11182014464,
                "cm1_2ns": 14445182976, "cm2_2ns": 27044872192, "cm3_2ns": 29863444480, "cm4_2ns": 36708548608,
                "q_2ns": 36809211904, "f_2ns": 36859543552, "end": 36909875200}


def layout(log_n, aliased=False, circ="zkevm"):
    """section -> (offset, width, rows) for N = 2^log_n, same packing rule as the real slab.

    aliased=True: the ONE-GPU layout.  The reference keeps all eleven sections apart in one 295 GB slab (and already lets
    dead regions double as scratch: starks.cpp:53,102-105); a 288 GB GPU cannot, but the sections are never all alive at once:
    the `_n` sections die with the last base-domain program (step3; only cm3_n lives on until its extension), so
        cm3_2ns                 lies over cm1_n + the head of cm2_n     (written by the step-3 LDE, when both are dead)
        cm4_2ns, q_2ns, f_2ns   lie inside tmpExp_n                     (written in step 4)
        "const_2ns"             (not a slab section in the reference; 234 x 2N words) lies over cm3_n .. head of tmpExp_n and is
                                re-extended from const_n per proof once cm3 is committed -- its tree stays resident
    which brings the slab from 4 609 N to 3 560 N words (239 GB at N = 2^23).  Offsets are in words, like the reference's."""
    o, out = 0, {}
    if not aliased:
        for name, w, ext in sections(circ):
            rows = 1 << (log_n + ext * CIRCUITS[circ]["ext"])
            out[name] = (o, w, rows)
            o += w * rows
        out["end"] = (o, 0, 0)
        return out
    assert circ == "zkevm"
    n = 1 << log_n
    width = {name: w for name, w, _ in SECTIONS}
    for name in ("cm1_n", "cm2_n", "cm3_n", "cm4_n", "tmpExp_n"):
        out[name] = (o, width[name], n); o += width[name] * n
    for name in ("cm1_2ns", "cm2_2ns"):
        out[name] = (o, width[name], 2 * n); o += width[name] * 2 * n
    out["end"] = (o, 0, 0)
    out["cm3_2ns"] = (0, width["cm3_2ns"], 2 * n)
    assert width["cm3_2ns"] * 2 * n <= out["cm3_n"][0]
    c2 = out["cm3_n"][0]
    out["const_2ns"] = (c2, N_CONST, 2 * n)
    q = c2 + N_CONST * 2 * n
    assert q >= out["tmpExp_n"][0]
    for name in ("cm4_2ns", "q_2ns", "f_2ns"):
        out[name] = (q, width[name], 2 * n); q += width[name] * 2 * n
    assert q <= out["cm1_2ns"][0]
    out["free"] = (q, 0, 0)                       # tmpExp_n's tail: dead from step 4 on
    return out


N_CONST = 234                                      # constant polynomials of the fork-9 zkEVM (highest const_2ns column of the step programs + 1)


def _check_real():
    L = layout(LOG_N_REAL)
    for k, v in REAL_OFFSETS.items():
        assert L[k][0] == v, (k, L[k][0], v)


def layout_window(log_n, world, circ="zkevm"):
    """per-rank slab of a proof row-sharded over `world` ranks: every section holds the rank's rows followed by the halo rows of the
    next rank (1 on the base domain, 2^ext on the extended one: the next-row distances of the step programs)"""
    ext = CIRCUITS[circ]["ext"]
    n, ne = (1 << log_n) // world, (1 << (log_n + ext)) // world
    o, out = 0, {}
    for name, w, e in sections(circ):
        rows = ne + (1 << ext) if e else n + 1
        out[name] = (o, w, rows); o += w * rows
    out["end"] = (o, 0, 0)
    return out


def relocate(codes, log_n, aliased=False, new_layout=None):
    """codes: converted zkEVM programs (real 2^23-row addressing) -> ([codes for N = 2^log_n], info); new_layout: another packing of the
    same sections (layout_window: the next-row operands keep the GLOBAL modulus 2^log_n and never wrap inside a window)"""
    _check_real()
    real, new = layout(LOG_N_REAL), (new_layout or layout(log_n, aliased))
    bases = sorted((real[name][0], name) for name, _, _ in SECTIONS)

    def move(off, stride):
        name = [n for b, n in bases if b <= off][-1]
        b, w, rows = real[name]
        col = off - b
        assert col < w and stride == w, (name, off, stride)
        return new[name][0] + col

    codes = [c.copy() for c in codes]
    info = {"n_const": 1, "n_const2": 1, "nch": 1, "npub": 1, "nev": 1}
    for code in codes:
        for _, _, kind, dim, q in R.operands(code):
            if kind == E.K_POLS:
                code[q] = move(int(code[q]), int(code[q + 1]))
            elif kind == E.K_POLS_PRIME:
                code[q] = move(int(code[q]), int(code[q + 3]))
            if kind in (E.K_POLS_PRIME, E.K_CONST_PRIME, E.K_CONST2NS_PRIME):
                n_real = int(code[q + 2])
                assert n_real in (1 << LOG_N_REAL, 1 << (LOG_N_REAL + 1)), n_real
                code[q + 2] = n_real >> (LOG_N_REAL - log_n)
            if kind in (E.K_CONST, E.K_CONST_PRIME):
                info["n_const"] = max(info["n_const"], int(code[q]) + 1)
            elif kind in (E.K_CONST2NS, E.K_CONST2NS_PRIME):
                info["n_const2"] = max(info["n_const2"], int(code[q]) + 1)
            elif kind == E.K_CHALLENGE:
                info["nch"] = max(info["nch"], int(code[q]) + 1)
            elif kind == E.K_PUBLIC:
                info["npub"] = max(info["npub"], int(code[q]) + 1)
            elif kind == E.K_EVAL:
                info["nev"] = max(info["nev"], int(code[q]) + 1)
    return codes, info


# ---------------------------------------------------------------------------------------------------------------------------
# COLUMN-MAJOR device sections (include/xlp_gl.h, XLP_EXPR_COLMAJOR).  The packing of the slab stays what `layout` says -- a section
# still occupies width x rows words at its offset -- but inside a section element (row, col) sits at col * rows + s(row), where the
# storage row s of an extended-domain row k = (q << ext) | r is r * (rows >> ext) + q (coset-major: the 2^ext cosets of the base
# domain one after the other), and s(row) = row on the base domain.
def to_colmajor(code, L, ext):
    """a program addressed for the row-major layout L (section -> (offset, width, rows)) -> the same program for the column-major
    form of L: XLP_K_POLS (offset, stride) -> (section offset + col * rows, rows); next-row operands on the extended domain step by
    next >> ext inside a coset of n >> ext rows"""
    secs = sorted((o, w, rows, name) for name, (o, w, rows) in L.items() if w and name != "const_2ns")
    code = code.copy()

    def move(off, stride):
        hit = [(o, w, rows, name) for o, w, rows, name in secs if o <= off < o + w * rows and w == stride]
        assert hit, (off, stride)
        o, w, rows, name = hit[-1]
        col = off - o
        assert col < w, (name, off)
        return o + col * rows, rows, name.endswith("_2ns")

    for _, _, kind, dim, q in R.operands(code):
        if kind == E.K_POLS:
            code[q], code[q + 1], _ = move(int(code[q]), int(code[q + 1]))
        elif kind == E.K_POLS_PRIME:
            off, rows, extended = move(int(code[q]), int(code[q + 3]))
            code[q], code[q + 3] = off, rows
            if extended:
                assert int(code[q + 1]) % (1 << ext) == 0 and int(code[q + 2]) % (1 << ext) == 0
                code[q + 1] = int(code[q + 1]) >> ext; code[q + 2] = int(code[q + 2]) >> ext
        elif kind == E.K_CONST2NS_PRIME:
            assert int(code[q + 1]) % (1 << ext) == 0 and int(code[q + 2]) % (1 << ext) == 0
            code[q + 1] = int(code[q + 1]) >> ext; code[q + 2] = int(code[q + 2]) >> ext
    return code


def dots_to_colmajor(dots, L):
    """the (kind, offset, stride, dim, group, coef) entries of a dot-columns plan (expr_opt.horner_to_dot) for the column-major form of L:
    offset -> the column's first word, stride -> the words between the columns of its section (constant entries keep their column index)"""
    secs = sorted((o, w, rows, name) for name, (o, w, rows) in L.items() if w and name != "const_2ns")
    out = []
    for kind, off, stride, dim, group, coef in dots:
        if kind == 0:
            o, w, rows, name = [t for t in secs if t[0] <= off < t[0] + t[1] * t[2] and t[1] == stride][-1]
            off, stride = o + (off - o) * rows, rows
        out.append((kind, off, stride, dim, group, coef))
    return out


def rows_to_storage(a, ext):
    """[rows][width] in natural row order -> [rows][width] in storage (coset-major) order; ext = 0: unchanged"""
    a = np.asarray(a)
    rows = a.shape[0]
    return a.reshape(rows >> ext, 1 << ext, -1).transpose(1, 0, 2).reshape(rows, -1)


def rows_from_storage(a, ext):
    a = np.asarray(a)
    rows = a.shape[0]
    return a.reshape(1 << ext, rows >> ext, -1).transpose(1, 0, 2).reshape(rows, -1)


def matrix_to_colmajor(a, ext):
    """row-major [rows][width], natural order -> flat column-major, coset-major storage order"""
    return np.ascontiguousarray(rows_to_storage(a, ext).T).reshape(-1)


def matrix_from_colmajor(flat, rows, width, ext):
    return rows_from_storage(np.asarray(flat).reshape(width, rows).T, ext)


def slab_to_colmajor(slab, L, ext):
    """the whole row-major slab of layout L -> its column-major form (same offsets, same size)"""
    out = np.array(slab, copy=True)
    for name, (o, w, rows) in L.items():
        if w and name != "const_2ns":
            out[o:o + w * rows] = matrix_to_colmajor(slab[o:o + w * rows].reshape(rows, w), ext if name.endswith("_2ns") else 0)
    return out


def slab_from_colmajor(slab, L, ext):
    out = np.array(slab, copy=True)
    for name, (o, w, rows) in L.items():
        if w and name != "const_2ns":
            out[o:o + w * rows] = matrix_from_colmajor(slab[o:o + w * rows], rows, w, ext if name.endswith("_2ns") else 0).reshape(-1)
    return out


STEPS = ("step2prev", "step3prev", "step3", "step42ns", "step52ns")


def scan_info(codes):
    """highest constant / challenge / public / evaluation index the programs name, + 1"""
    info = {"n_const": 1, "n_const2": 1, "nch": 1, "npub": 1, "nev": 1}
    for code in codes:
        for _, _, kind, dim, q in R.operands(code):
            if kind in (E.K_CONST, E.K_CONST_PRIME):
                info["n_const"] = max(info["n_const"], int(code[q]) + 1)
            elif kind in (E.K_CONST2NS, E.K_CONST2NS_PRIME):
                info["n_const2"] = max(info["n_const2"], int(code[q]) + 1)
            elif kind == E.K_CHALLENGE:
                info["nch"] = max(info["nch"], int(code[q]) + 1)
            elif kind == E.K_PUBLIC:
                info["npub"] = max(info["npub"], int(code[q]) + 1)
            elif kind == E.K_EVAL:
                info["nev"] = max(info["nev"], int(code[q]) + 1)
    return info


def load(log_n, aliased=False, circ="zkevm", new_layout=None):
    import os
    steps = CIRCUITS[circ]["steps"]
    progs = [R.read_program(os.path.join(R.REF_DIR, f"{circ}.{s}.xlpexpr")) for s in steps]
    if circ == "zkevm":
        codes, info = relocate([p[0] for p in progs], log_n, aliased, new_layout)
    else:                                           # small STARKs run at their real geometry: the programs as converted
        assert log_n == CIRCUITS[circ]["log_n"] and not aliased
        codes = [p[0] for p in progs]
        info = scan_info(codes)
        L = layout(log_n, False, circ)
        for code in codes:                          # every column operand must fall inside the section its stride names
            for _, _, kind, dim, q in R.operands(code):
                if kind in (E.K_POLS, E.K_POLS_PRIME):
                    off, stride = int(code[q]), int(code[q + 1] if kind == E.K_POLS else code[q + 3])
                    assert any(w == stride and o <= off < o + w for name, (o, w, rows) in L.items() if w), (circ, off, stride)
    return {s: (c, p[1], p[2]) for s, c, p in zip(steps, codes, progs)}, info


NCH = 8                                             # challenges of a STARK proof (starks.cpp: 0..7)


def load_opt(log_n, aliased=False, circ="zkevm", new_layout=None):
    """every step with its Horner chains turned into dot products (xlayer_prover_amd/expr_opt.py):
    step -> ((row code, ntmp1, ntmp3), table Program, number of extended-challenge entries, statistics)"""
    from xlayer_prover_amd import expr_opt
    progs, _ = load(log_n, aliased, circ, new_layout)
    res = {}
    for step, (code, n1, n3) in progs.items():
        row, tab, n_ext, stats = expr_opt.horner_to_dot(code, n1, n3, NCH)
        res[step] = ((row.code, row.ntmp1, row.ntmp3), tab, n_ext, stats)
    return res


def load_fri_dots(log_n, aliased=False, circ="zkevm", new_layout=None):
    """step52ns as (table Program, dot-columns groups, 12-instruction row Program, n_ext): xlayer_prover_amd/expr_opt.py with
    its big sums handed to xlp_dev_dot_columns"""
    from xlayer_prover_amd import expr_opt
    progs, _ = load(log_n, aliased, circ, new_layout)
    code, n1, n3 = progs["step52ns"]
    row, tab, n_ext, stats, dots = expr_opt.horner_to_dot(code, n1, n3, NCH, outsource_min_terms=8)
    assert dots is not None
    return tab, dots, row, n_ext


# ---------------------------------------------------------------------------------------------------------------------------
# A starkinfo.json for these geometries (src/starkpil/stark_info.cpp schema).  The reference's files are in the missing
# archive; what the step programs fix is reproduced exactly (section widths and offsets, the evaluation map read off
# step52ns, nConstants, nPublics, qDeg), what only the PIL would tell is chosen here and shared by every driver
# (tools/bench_zkevm_genproof.py, host/starks.hpp through the bundle): which tmpExp_n columns are the plookup tables and the
# grand-product numerators / denominators.
def _tmpexp_dests(code, L):
    """sorted first columns of the dimension-3 tmpExp_n polynomials a base-domain program writes"""
    o, w, rows = L["tmpExp_n"]
    dst = set()
    ops = list(R.operands(code))
    for i in range(0, len(ops), 3):
        _, _, kind, dim, q = ops[i]
        if kind == E.K_POLS and dim == 3 and o <= int(code[q]) < o + w * rows and int(code[q + 1]) == w:
            dst.add(int(code[q]) - o)
    return sorted(dst)


def lookup_picks(circ, progs=None, L=None):
    """-> (lookups, zs).  Lookup k reads its table t from the tmpExp_n polynomial (3 columns from) lookups[k] -- one of the
    extension-field expressions step2prev computes, as a real plookup's compressed table is -- and writes h1 / h2 to cm2_n columns
    [6k, 6k+3) / [6k+3, 6k+6); f is a synthetic selection of rows of t (xlp_dev_gather_rows: on a synthetic witness "f in t" has
    to be made).  Grand product k reads num / den from the tmpExp_n polynomials zs[k] = (num column, den column), two of the
    expressions step3prev computes, and writes z to cm3_n columns [3k, 3k+3)."""
    C = CIRCUITS[circ]
    if progs is None:
        progs, _ = load(C["log_n"] if circ != "zkevm" else 14, circ == "zkevm", circ)
        L = layout(C["log_n"] if circ != "zkevm" else 14, circ == "zkevm", circ)
    d2 = _tmpexp_dests(progs["step2prev"][0], L) if "step2prev" in progs else []
    d3 = _tmpexp_dests(progs["step3prev"][0], L)
    assert len(d3) >= 2 and (d2 or not C["lookups"])
    lookups = [d2[(2 * k + 1) % len(d2)] for k in range(C["lookups"])]
    zs = [(d3[(2 * k) % (len(d3) & ~1)], d3[(2 * k + 1) % (len(d3) & ~1)]) for k in range(C["zs"])]
    return lookups, zs


def evmap_of(code52, L, secs, nconst):
    """[(kind 'cm' | 'const', section name | None, column, dim, prime)] in evals order, from the FRI-polynomial program: each
    `column - evals[k]` difference, and whether its chain is finally multiplied by xDivXSubXi or xDivXSubWXi"""
    ents, tags, prime = {}, {}, {}
    ops = list(R.operands(code52))
    for i in range(0, len(ops), 3):
        slots = [(kind, dim, [int(v) for v in code52[q:q + E.KIND_PARAMS[kind]]]) for _, _, kind, dim, q in ops[i:i + 3]]
        op = int(code52[ops[i][4] - 1]) & 255
        (dk, dd, dp), (ak, ad, apar), (bk, bd, bpar) = slots
        srcs = [(ak, ad, apar)] + ([(bk, bd, bpar)] if bk != E.K_NONE else [])
        s = set()
        for kind, dim, prm in srcs:
            if kind == E.K_TMP3:
                s |= tags.get(prm[0], set())
        if op == E.SUB and bk == E.K_EVAL and ak in (E.K_POLS, E.K_CONST2NS):
            if ak == E.K_POLS:
                off = apar[0]
                name = [sn for sn, _, ext in secs if ext and L[sn][1] and L[sn][0] <= off < L[sn][0] + L[sn][1] * L[sn][2]][-1]
                ents[bpar[0]] = ("cm", name, off - L[name][0], ad)
            else:
                ents[bpar[0]] = ("const", None, apar[0], 1)
            s.add(bpar[0])
        for kind, dim, prm in srcs:
            if kind in (E.K_XDIVXSUBXI, E.K_XDIVXSUBWXI):
                for k in s:
                    prime[k] = kind == E.K_XDIVXSUBWXI
                s = set()
        if dk == E.K_TMP3:
            tags[dp[0]] = s
    assert set(prime) == set(ents) and sorted(ents) == list(range(len(ents)))
    return [ents[k] + (prime[k],) for k in sorted(ents)]


def next_row_stores(code, L):
    """[(section, first column, columns, next distance)] of a program's stores to NEXT rows (zkEVM step3prev / step3 have a few): in a
    row-sharded run the rows they write past a rank's last row belong to the next rank (xlayer_prover_amd/shard.py halo_writeback)"""
    out = set()
    for op, d, a, b in E.decode(code):
        if d.kind == E.K_POLS_PRIME:
            off, nxt, _, stride = d.params
            name = [nm for nm, (o, w, r) in L.items() if w and o <= off < o + w * r and w == stride][-1]
            out.add((name, int(off - L[name][0]), int(d.dim), int(nxt)))
    return sorted(out)


def synth_starkinfo(circ, log_n, aliased=False, fri_steps=None, queries=None, world=0):
    """world > 0: the per-rank WINDOW layout of a proof row-sharded over `world` ranks (layout_window): every section holds the rank's
    N / world rows followed by the halo rows of the next rank; offsets, mapDeg and mapTotalN describe that window slab, nBits / nBitsExt
    stay the global ones.  host/starks.hpp runs such a starkinfo in its sharded mode (world must equal the number of ranks)."""
    C = CIRCUITS[circ]
    ext = C["ext"]
    assert not (world and aliased)
    L = layout_window(log_n, world, circ) if world else layout(log_n, aliased, circ)
    secs = sections(circ)
    progs, info = load(log_n, aliased, circ, L if world else None)
    nconst = max(info["n_const"], info["n_const2"])
    W = {name: w for name, w, _ in secs}
    lne = log_n + ext
    if fri_steps is None:
        fri_steps = C["fri"] if C["fri"] and log_n == C["log_n"] else [lne] + [b for b in (lne - 5, lne - 10, lne - 14, lne - 18) if b >= 3]
    lookups, zs = lookup_picks(circ, progs, L)
    evm = evmap_of(progs["step52ns"][0], L, secs, nconst)
    # ---- polynomials: base sections first (cm_n order = commit order), then their extended twins, then the tmpExp operands
    var, cm_n = [], []
    dims3 = {}                                               # the picks above decide where h1 / h2 / z go; everything else is a column

    def cols_of(base):
        """[(pos, dim)] of a base section"""
        w, twin = W[base], base.replace("_n", "_2ns")
        out, c = [], 0
        if base == "cm2_n":
            for k in range(len(lookups)):
                out += [(6 * k, 3), (6 * k + 3, 3)]
            c = 6 * len(lookups)
        elif base == "cm3_n":
            out += [(3 * k, 3) for k in range(len(zs))]
            c = 3 * len(zs)
        elif base == "cm4_n":
            out += [(3 * p, 3) for p in range(w // 3)]
            c = 3 * (w // 3)
        while c < w:
            d = 3 if (twin, c) in dims3 and c + 3 <= w else 1
            out.append((c, d)); c += d
        return out
    counts = {}
    for base in ("cm1_n", "cm2_n", "cm3_n", "cm4_n"):
        cs = cols_of(base)
        counts[base] = len(cs)
        for pos, d in cs:
            cm_n.append(len(var)); var.append({"section": base, "dim": d, "sectionPos": pos})
    cm_2ns, twin_id = [], {}
    for i in cm_n:
        v = var[i]
        sec2 = v["section"].replace("_n", "_2ns")
        twin_id[(sec2, v["sectionPos"], v["dim"])] = len(var)
        cm_2ns.append(len(var)); var.append({"section": sec2, "dim": v["dim"], "sectionPos": v["sectionPos"]})
    # the evaluation map names (column, dimension) views of the extended sections as the PIL really lays them out; where that differs
    # from the views the picks above created (varPolMap entries are views, they may overlap), the view is added
    for kind, name, col, dim, prime in evm:
        if kind == "cm" and (name, col, dim) not in twin_id:
            twin_id[(name, col, dim)] = len(var)
            cm_2ns.append(len(var)); var.append({"section": name, "dim": dim, "sectionPos": col})
    qs = [twin_id[("cm4_2ns", 3 * p, 3)] for p in range(W["cm4_n"] // 3)]
    exp2pol, pu, pe = {}, [], []

    def tmp_pol(pos):
        eid = 1000 + len(exp2pol)
        exp2pol[str(eid)] = len(var); var.append({"section": "tmpExp_n", "dim": 3, "sectionPos": pos})
        return eid
    for k, c in enumerate(lookups):
        t = tmp_pol(c)
        pu.append({"tExpId": t, "fExpId": t, "h1Id": cm_n[counts["cm1_n"] + 2 * k], "h2Id": cm_n[counts["cm1_n"] + 2 * k + 1], "zId": 0, "c1Id": 0, "numId": 0, "denId": 0, "c2Id": 0})
    z0 = counts["cm1_n"] + counts["cm2_n"]
    for k, (cn, cd) in enumerate(zs):
        num, den = tmp_pol(cn), tmp_pol(cd)
        ctx = {"zId": cm_n[z0 + k], "numId": num, "denId": den, "c1Id": 0, "c2Id": 0}
        if k < len(pu):
            pu[k].update(ctx)
        else:
            pe.append({"tExpId": 0, "fExpId": 0, **ctx})
    ev = []
    for kind, name, col, dim, prime in evm:
        if kind == "const":
            ev.append({"type": "const", "id": col, "prime": prime})
        elif name == "cm4_2ns" and dim == 3 and col % 3 == 0:
            ev.append({"type": "q", "id": col // 3, "prime": prime})
        else:
            ev.append({"type": "cm", "id": cm_2ns.index(twin_id[(name, col, dim)]), "prime": prime})
    order = [name for name, _, _ in secs]
    rows = {name: (L[name][2] if world else 1 << (log_n + (ext if e else 0))) for name, _, e in secs}
    step = {"tmpUsed": 0, "first": [], "i": [], "last": []}
    j = {"starkStruct": {"nBits": log_n, "nBitsExt": lne, "nQueries": queries or C["queries"], "verificationHashType": C.get("hash", "GL"), "steps": [{"nBits": b} for b in fri_steps]},
         "mapTotalN": L["end"][0], "nConstants": nconst, "nPublics": info["npub"], "nCm1": counts["cm1_n"], "nCm2": counts["cm2_n"], "nCm3": counts["cm3_n"],
         "nCm4": counts["cm4_n"], "friExpId": 0, "nExps": len(exp2pol), "qDim": 3, "qDeg": W["cm4_n"] // 3,
         "mapDeg": rows, "mapOffsets": {name: L[name][0] for name in order}, "mapSections": {name: [] for name in order}, "mapSectionsN": {name: W[name] for name in order},
         "mapSectionsN1": {name: W[name] for name in order}, "mapSectionsN3": {name: 0 for name in order}, "varPolMap": var, "qs": qs, "cm_n": cm_n, "cm_2ns": cm_2ns,
         "peCtx": pe, "puCtx": pu, "ciCtx": [], "evMap": ev, "step2prev": step, "step3prev": step, "step3": step, "step42ns": step, "step52ns": step,
         "exps_n": [], "q_2ns": [], "cm4_n": [], "cm4_2ns": [], "tmpExp_n": [], "exp2pol": exp2pol,
         # not in the reference's schema: where the extended constants live when they share the slab (one-GPU aliased layout), and that
         # the lookups of this STARK run on a synthetic witness (f is then selected from t, see xlp_dev_gather_rows)
         "xlp": {"aliased": bool(aliased), "const2nsOffset": L["const_2ns"][0] if aliased else None, "circuit": circ, "steps": list(C["steps"])}}
    if world:
        j["xlp"].update({"world": world, "haloN": 1, "haloExt": 1 << ext})
    return j


def program_io_words(code):
    """(distinct words read per row, distinct words written per row) of a step program: every column / table word it names once --
    the algorithmic HBM traffic of one evaluation (temporaries, challenges and constants of the program itself are not traffic)"""
    rd, wr = set(), set()
    ops = list(R.operands(code))
    for i in range(0, len(ops), 3):
        for s, (_, _, kind, dim, q) in enumerate(ops[i:i + 3]):
            prm = [int(v) for v in code[q:q + E.KIND_PARAMS[kind]]]
            if kind in (E.K_POLS, E.K_POLS_PRIME):
                keys = [("p", prm[0] + c, kind == E.K_POLS_PRIME) for c in range(dim)]
            elif kind in (E.K_CONST, E.K_CONST_PRIME, E.K_CONST2NS, E.K_CONST2NS_PRIME):
                keys = [("c", kind, prm[0])]
            elif kind in (E.K_X_N, E.K_X_2NS, E.K_ZI):
                keys = [("x", kind)]
            elif kind in (E.K_XDIVXSUBXI, E.K_XDIVXSUBWXI, E.K_Q_2NS, E.K_F_2NS):
                keys = [("e", kind, c) for c in range(3)]
            else:
                continue
            (wr if s == 0 else rd).update(keys)
    return len(rd), len(wr)
