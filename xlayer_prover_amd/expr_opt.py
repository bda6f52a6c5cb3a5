"""Program-level optimisation for the constraint-expression VM: Horner chains -> dot products.

The generated step programs of a STARK (src/starkpil/zkevm/chelpers/zkevm.chelpers.step*.parser.cpp) contain long Horner
chains in a challenge v over COLUMNS:

    acc = (...((Y0 v + Y1) v + Y2) v + ...) ,   Y_k = col_k  or  col_k - evals[e_k]

-- the FRI polynomial (step52ns: 2 990 of them for the zkEVM, 75 % of the program) and the plookup / permutation
compressions `a0 + u (a1 + u (a2 + ...))` of step3prev / step3 / step42ns (1 291 more).  Each link is an
Fp^3 x Fp^3 product per row although most columns are base field.  Since v and the evaluations do not depend on the
row, the same value is

    acc = sum_k (+-) col_k * v^(p_k)  -  D ,      D = sum_k (+-) evals[e_k] * v^(p_k) ,

with the powers of v and D computed ONCE per proof: a base-field column then costs 3 base multiplications instead of a
9-multiplication (Karatsuba 6) extension product.  Field arithmetic is exact, so the rewritten program produces the same
canonical values as the original.

`horner_to_dot(code, ntmp1, ntmp3, nch)` returns
    row program   -- reads the table as CHALLENGE operands with indices >= nch,
    table program -- row independent; run it on ONE row with `pols` = the extended challenge buffer
                     (3 * n_ext words, the first 3 * nch hold the proof's challenges); it fills entries nch .. n_ext-1,
    n_ext, statistics.
Both are ordinary VM programs: nothing changes in the interpreter or the code generator.

With `outsource_min_terms=k` the plain-column part (>= k terms) of every sum of columns, temporaries and constants becomes one GROUP of
`xlp_dev_dot_columns` (csrc/polinomial.hip: sections read once, coalesced, products accumulated unreduced) instead of VM
instructions; a fifth value describes the groups (`{"ngroups", "entries": [(column operand, group, coefficient index)]}`)
and the row program then reads group g as the Fp^3 column 3g of ITS `pols` buffer = the kernel's output (rows x 3 ngroups).
This is only offered when the row program is left without any read of a real column (the FRI-polynomial programs are:
zkEVM step52ns becomes 3 groups of 2 992 entries + 12 instructions); otherwise the fifth value is None and the inline form
is returned.

How: every extension temporary is tracked symbolically as  base v^b + sum (+-) col v^p - sum (+-) evals[e] v^p  for as long
as the program only shifts it (multiplies by the chain's challenge) or adds / subtracts columns, `col - evals[e]` values
values, constants, other temporaries or other such values; base-field temporaries that are plain copies of a column stand
for that column.  Only when another instruction needs the value is it emitted, in dot-product form.  A symbolic value
never outlives a write to something it reads: before a column is written every symbolic value is evaluated (the programs
of steps 2-3 write tmpExp / cm3 columns), before a temporary is overwritten those that read it are.  Instructions left
without a reader are dropped."""
from . import expr as E

_COL = (E.K_POLS, E.K_POLS_PRIME, E.K_CONST, E.K_CONST_PRIME, E.K_CONST2NS, E.K_CONST2NS_PRIME)


class _Lin:
    """base * v^bpw + sum sign * X * v^pw - sum sign * evals[e] * v^pw   (v = challenge c; base = the slot's own register)"""
    __slots__ = ("c", "terms", "evals", "bpw")

    def __init__(self, c=None, terms=(), evals=(), bpw=None):
        self.c, self.terms, self.evals, self.bpw = c, list(terms), list(evals), bpw

    def shifted(self, c):
        return _Lin(c, [(x, p + 1, s) for x, p, s in self.terms], [(e, p + 1, s) for e, p, s in self.evals], None if self.bpw is None else self.bpw + 1)

    def negated(self):
        assert self.bpw is None
        return _Lin(self.c, [(x, p, -s) for x, p, s in self.terms], [(e, p, -s) for e, p, s in self.evals])

    def plus(self, o):
        assert o.bpw is None
        return _Lin(self.c if self.c is not None else o.c, self.terms + o.terms, self.evals + o.evals, self.bpw)

    def compatible(self, o):
        return self.c is None or o.c is None or self.c == o.c


def horner_to_dot(code, ntmp1, ntmp3, nch, outsource_min_terms=0, _ngroups=None, hoist=True):
    if outsource_min_terms and _ngroups is None:          # first pass counts the dot-product groups (their row stride)
        probe = horner_to_dot(code, ntmp1, ntmp3, nch, outsource_min_terms, _ngroups=1, hoist=False)
        if probe[4] is None:
            return horner_to_dot(code, ntmp1, ntmp3, nch, hoist=hoist) + (None,)
        return horner_to_dot(code, ntmp1, ntmp3, nch, outsource_min_terms, _ngroups=probe[4]["ngroups"], hoist=hoist)
    out, tab = E.Program(), E.Program()
    scratch = E.tmp3(ntmp3)                                  # one more extension temporary for the products
    state = {}                                               # tmp3 slot -> _Lin (value not in its register yet)
    alias = {}                                               # tmp1 slot -> the column it is a copy of
    powers = {}                                              # (c, j) -> table index of v_c^j
    n_ext = [nch]
    stats = {"dot_products": 0, "ext_mul_before": 0, "ext_mul_after": 0}
    dots = {"ngroups": 0, "entries": []}                     # big all-column sums handed to xlp_dev_dot_columns
    negs = {}

    def entry():
        n_ext[0] += 1
        return n_ext[0] - 1

    def tab_pols(idx): return E.pols(3 * idx, 3, 3)

    def need_power(c, j):
        for k in range(2, j + 1):
            if (c, k) not in powers:
                prev = E.challenge(c) if k == 2 else tab_pols(powers[(c, k - 1)])
                idx = entry()
                tab.mul(tab_pols(idx), prev, E.challenge(c))
                powers[(c, k)] = idx

    def power(c, j):                                         # operand of the row program
        need_power(c, j)
        return E.challenge(c if j == 1 else powers[(c, j)])

    def tab_power(c, j):                                     # operand of the table program
        need_power(c, j)
        return E.challenge(c) if j == 1 else tab_pols(powers[(c, j)])

    alphas = {}

    def alpha_times(idx):
        """table entry holding k * a for the entry / challenge k = idx (a: the generator of Fp^3, a^3 = a + 1):
        (k0, k1, k2) a = (k2, k0 + k2, k1)"""
        if idx not in alphas:
            src = idx
            if idx < nch:                                    # a challenge: its components are only addressable from the table
                src = entry()
                tab.copy(tab_pols(src), E.challenge(idx))
            j = entry()
            w = lambda e, c: E.pols(3 * e + c, 3, 1)
            tab.copy(w(j, 0), w(src, 2)).add(w(j, 1), w(src, 0), w(src, 2)).copy(w(j, 2), w(src, 1))
            alphas[idx] = j
        return alphas[idx]

    def emit_eval_constant(lin, dst):
        """dst -= D,  D = sum (+-) evals[e] * v^p over the evaluation terms with p >= 1: one table entry"""
        hi_e = [t for t in lin.evals if t[1] >= 1]
        if not hi_e:
            return
        t0, t1 = E.tmp3(0), E.tmp3(1)
        for k, (e, p, s) in enumerate(hi_e):
            if k == 0:
                tab.mul(t0, E.evaluation(e), tab_power(lin.c, p))
                if s < 0:
                    tab.sub(t0, E.number(0), t0)
            else:
                tab.mul(t1, E.evaluation(e), tab_power(lin.c, p))
                (tab.add if s > 0 else tab.sub)(t0, t0, t1)
        idx = entry()
        tab.copy(tab_pols(idx), t0)
        out.sub(dst, dst, E.challenge(idx))

    def materialise(slot):
        if slot not in state:
            return
        barrier(E.K_TMP3, slot, exclude=slot)                # others that read this register's present value go first
        lin = state.pop(slot)
        dst = E.tmp3(slot)
        plain = [t for t in lin.terms if t[0].kind in (E.K_POLS, E.K_CONST, E.K_CONST2NS) and (t[0].dim == 1 or t[0].kind == E.K_POLS)]
        if outsource_min_terms and len(plain) >= outsource_min_terms and \
                all(x.kind in (E.K_POLS, E.K_CONST, E.K_CONST2NS, E.K_TMP1, E.K_TMP3, E.K_NUMBER) for x, _, _ in lin.terms):
            # the sum over the plain columns is one group of the dot-columns kernel; its result is column g of the buffer the
            # row program gets as `pols`.  Terms that are temporaries or constants (e.g. the recursive STARKs add an already
            # evaluated partial sum) stay in the row program.
            rest = [t for t in lin.terms if not any(t is q for q in plain)]
            g = dots["ngroups"]; dots["ngroups"] += 1
            for x, p, sg in plain:
                if p >= 2:
                    need_power(lin.c, p)
                idx = one_entry() if p == 0 else (lin.c if p == 1 else powers[(lin.c, p)])
                if sg < 0:
                    if idx not in negs:
                        negs[idx] = entry()
                        tab.sub(E.tmp3(0), E.challenge(0), E.challenge(0)).sub(tab_pols(negs[idx]), E.tmp3(0),
                                                                                E.challenge(idx) if idx < nch else tab_pols(idx))
                    idx = negs[idx]
                if x.dim == 1:
                    dots["entries"].append((x, g, idx))
                else:
                    # k * (x0 + x1 a + x2 a^2) = x0 * k + x1 * (k a) + x2 * (k a^2): three base-field entries, so the kernel's
                    # cheap unreduced path serves extension columns too
                    ka = alpha_times(idx)
                    kaa = alpha_times(ka)
                    for w, ci in enumerate((idx, ka, kaa)):
                        xw = E.pols(x.params[0] + w, x.params[1], 1) if x.kind == E.K_POLS else None
                        assert xw is not None, "extension column outside pols"
                        dots["entries"].append((xw, g, ci))
            res = E.pols(3 * g, 3 * (_ngroups or 1), 3)
            if lin.bpw is None:
                out.copy(dst, res)
            else:                                            # base * v^b + the sum
                if lin.bpw >= 1:
                    out.mul(dst, dst, power(lin.c, lin.bpw))
                out.add(dst, dst, res)
            for x, p, sg in rest:
                if p == 0:
                    (out.add if sg > 0 else out.sub)(dst, dst, x)
                else:
                    out.mul(scratch, x, power(lin.c, p))
                    (out.add if sg > 0 else out.sub)(dst, dst, scratch)
            for e, p, sg in [t for t in lin.evals if t[1] == 0]:
                (out.sub if sg > 0 else out.add)(dst, dst, E.evaluation(e))
            emit_eval_constant(lin, dst)
            stats["dot_products"] += 1
            return
        hi_t = sorted([t for t in lin.terms if t[1] >= 1], key=lambda t: -t[2])          # positive terms first
        lo_t = sorted([t for t in lin.terms if t[1] == 0], key=lambda t: -t[2])
        hi_e = [t for t in lin.evals if t[1] >= 1]
        lo_e = [t for t in lin.evals if t[1] == 0]
        started = False
        if lin.bpw is not None:
            if lin.bpw >= 1:
                out.mul(dst, dst, power(lin.c, lin.bpw))
            started = True
        for x, p, s in hi_t:
            if not started:
                out.mul(dst, x, power(lin.c, p)); started = True
                if s < 0:
                    out.sub(dst, E.number(0), dst)
            else:
                out.mul(scratch, x, power(lin.c, p))
                (out.add if s > 0 else out.sub)(dst, dst, scratch)
        for x, p, s in lo_t:
            if not started:
                pos_e = [t for t in lo_e if t[2] > 0]
                if s > 0 and pos_e:                          # the original `col - evals[e]` form
                    lo_e.remove(pos_e[0])
                    out.sub(dst, x, E.evaluation(pos_e[0][0]))
                else:                                        # rare: widen through a multiplication by one
                    out.mul(dst, x, E.challenge(one_entry()))
                    if s < 0:
                        out.sub(dst, E.number(0), dst)
                started = True
            else:
                (out.add if s > 0 else out.sub)(dst, dst, x)
        assert started, "empty symbolic value"
        for e, p, s in lo_e:
            (out.sub if s > 0 else out.add)(dst, dst, E.evaluation(e))
        emit_eval_constant(lin, dst)
        if len(hi_t) > 1:
            stats["dot_products"] += 1

    one = []

    def one_entry():                                         # (1, 0, 0) in the table
        if not one:
            one.append(entry())
            tab.sub(E.tmp3(0), E.challenge(0), E.challenge(0)).add(tab_pols(one[0]), E.number(1), E.tmp3(0))
        return one[0]

    def is_t3(o): return o is not None and o.kind == E.K_TMP3
    def is_ch(o): return o is not None and o.kind == E.K_CHALLENGE and o.params[0] < nch

    def column(o):
        if o is None:
            return None
        if o.kind in _COL:
            return o
        if o.kind == E.K_TMP1:
            return alias.get(o.params[0])
        return None

    def lin_of(o, d_slot):
        """symbolic value of a source operand, or None if it cannot be part of a chain"""
        if o is None:
            return None
        if is_t3(o):
            k = o.params[0]
            if k in state:
                return state[k]
            return _Lin(bpw=0) if k == d_slot else _Lin(None, [(o, 0, 1)])     # a real value: the base / a term
        x = column(o)
        if x is not None:
            return _Lin(None, [(x, 0, 1)])
        if o.kind in (E.K_TMP1, E.K_NUMBER):
            return _Lin(None, [(o, 0, 1)])
        return None

    def refs(lin):
        return {(x.kind, x.params[0]) for x, _, _ in lin.terms if x.kind in (E.K_TMP1, E.K_TMP3)}

    def barrier(kind, slot, exclude=None):
        """the register (kind, slot) is about to change: evaluate every symbolic value that still reads it"""
        for k in list(state):
            if k != exclude and k in state and (kind, slot) in refs(state[k]):
                materialise(k)

    def flush_all():
        for k in list(state):
            materialise(k)

    def acceptable(lin, slot):
        """may `lin` become the symbolic value of tmp3 `slot`?  It must not read, as a term, the register it will be written to,
        neither directly nor through the symbolic values of the registers it reads (evaluation order would be circular)."""
        todo, seen = [k for kind, k in refs(lin) if kind == E.K_TMP3], set()
        while todo:
            k = todo.pop()
            if k == slot:
                return False
            if k in seen:
                continue
            seen.add(k)
            if k in state:
                todo += [j for kind, j in refs(state[k]) if kind == E.K_TMP3]
        return True

    for op, d, a, b in E.decode(code):
        if op == E.MUL and a.dim == 3 and b.dim == 3:
            stats["ext_mul_before"] += 1
        if is_t3(d):
            s = d.params[0]
            if op == E.SUB and column(a) is not None and b.kind == E.K_EVAL:
                state[s] = _Lin(None, [(column(a), 0, 1)], [(b.params[0], 0, 1)])
                continue
            if op == E.MUL and (is_ch(a) or is_ch(b)):
                ch, x = (a, b) if is_ch(a) else (b, a)
                c = ch.params[0]
                lin = lin_of(x, s)
                if lin is not None and is_t3(x) and x.params[0] in state and \
                        ((lin.bpw is not None and x.params[0] != s) or lin.c not in (None, c)):
                    materialise(x.params[0])                 # base in another register, or a chain in another challenge:
                    lin = lin_of(x, s)                       # take its real value
                if lin is not None and lin.c in (None, c) and acceptable(lin, s):
                    state[s] = lin.shifted(c)
                    continue
            if op in (E.ADD, E.SUB):
                la, lb = lin_of(a, s), lin_of(b, s)
                one_is_chain = (is_t3(a) and a.params[0] in state) or (is_t3(b) and b.params[0] in state)
                if la is not None and lb is not None and one_is_chain and la.compatible(lb):
                    if op == E.ADD and la.bpw is None and lb.bpw is not None:
                        la, lb, a, b = lb, la, b, a                      # x + acc == acc + x
                    if lb.bpw is None and (la.bpw is None or (is_t3(a) and a.params[0] == s)):
                        cand = la.plus(lb if op == E.ADD else lb.negated())
                        if acceptable(cand, s):
                            state[s] = cand
                            continue
        # anything else: its tmp3 sources must hold real values; its destination becomes a real value
        if d.kind in (E.K_POLS, E.K_POLS_PRIME):
            flush_all()                                      # a column is about to change: nothing symbolic may outlive it
            alias.clear()
        for o in (a, b):
            if is_t3(o):
                materialise(o.params[0])
        if d.kind in (E.K_TMP1, E.K_TMP3):
            barrier(d.kind, d.params[0])
        if is_t3(d):
            state.pop(d.params[0], None)
        if d.kind == E.K_TMP1:
            alias.pop(d.params[0], None)
            if op == E.COPY and a.kind in _COL:
                alias[d.params[0]] = a
        out.emit(op, d, a, b)
    out = _drop_dead(out)
    for op, d, a, b in E.decode(out.code):
        if op == E.MUL and a.dim == 3 and b.dim == 3:
            stats["ext_mul_after"] += 1
    out.ntmp1 = max(out.ntmp1, ntmp1)
    out.ntmp3 = max(out.ntmp3, ntmp3 + 1)
    tab.ntmp3 = max(tab.ntmp3, 2)
    if hoist:
        out = hoist_sum_terms(out)                           # the terms of the sums go where their operands are produced
    if outsource_min_terms:
        # the row program's `pols` will be the dot-columns result: it must not read real columns any more
        clean = dots["ngroups"] and all(o is None or o.kind not in _COL or (o.kind == E.K_POLS and o.params[1] == 3 * (_ngroups or 1))
                                        for _, d, a, b in E.decode(out.code) for o in (d, a, b))
        return out, tab, n_ext[0], stats, (dots if clean else None)
    return out, tab, n_ext[0], stats


def _drop_dead(prog):
    """remove instructions whose destination is a temporary nobody reads (backward liveness over the straight-line code)"""
    ins = list(E.decode(prog.code))
    live, keep = set(), [False] * len(ins)
    for i in range(len(ins) - 1, -1, -1):
        op, d, a, b = ins[i]
        key = (d.kind, d.params[0]) if d.kind in (E.K_TMP1, E.K_TMP3) else None
        if key is not None and key not in live:
            continue
        keep[i] = True
        if key is not None:
            live.discard(key)
        for o in (a, b):
            if o is not None and o.kind in (E.K_TMP1, E.K_TMP3):
                live.add((o.kind, o.params[0]))
    res = E.Program()
    for k, (op, d, a, b) in zip(keep, ins):
        if k:
            res.emit(op, d, a, b)
    return res


def hoist_sum_terms(prog, min_distance=64, min_terms=8):
    """Move the terms of the big sums next to where their operands are produced.

    After horner_to_dot a sum over temporaries is emitted where its value is needed: a run of
        mul S, x_k, K_k ; add|sub D, D, S          (S: the scratch register, K_k: a row-independent table entry)
    whose x_k were often computed thousands of instructions earlier (zkEVM step42ns: 1 600 temporaries wait for the final sum).
    In the ahead-of-time form every such value crosses segment boundaries through HBM (one store + one load per row).  Field
    addition is associative and commutative and exact, so each pair may run right after the last write of x_k instead,
    accumulating into a fresh register A (zeroed at the start of the program); the run keeps one `add D, D, A` in place of the
    pairs that left.  Same value in D bit for bit; x_k dies where it is born.  Returns a new Program (prog if nothing moved)."""
    ins = list(E.decode(prog.code))
    n = len(ins)
    tkey = lambda o: (o.kind, o.params[0]) if o is not None and o.kind in (E.K_TMP1, E.K_TMP3) else None
    # position of the last write of every temporary before each instruction: one forward pass
    last_write = {}                                           # temp -> sorted positions where it is written
    reads_after = {}                                          # temp -> sorted positions where it is read (for the scratch check)
    for i, (op, d, a, b) in enumerate(ins):
        for o in (a, b):
            k = tkey(o)
            if k is not None:
                reads_after.setdefault(k, []).append(i)
        k = tkey(d)
        if k is not None:
            last_write.setdefault(k, []).append(i)
    import bisect

    def last_def_before(k, i):
        w = last_write.get(k, [])
        j = bisect.bisect_left(w, i) - 1
        return w[j] if j >= 0 else None

    def is_pair(i):
        if i + 1 >= n:
            return None
        op, d, a, b = ins[i]
        op2, d2, a2, b2 = ins[i + 1]
        if op != E.MUL or d.kind != E.K_TMP3 or op2 not in (E.ADD, E.SUB) or d2.kind != E.K_TMP3:
            return None
        S, D = tkey(d), tkey(d2)
        if S == D or tkey(a2) != D or tkey(b2) != S:
            return None
        x, K = (a, b) if tkey(a) is not None else (b, a)
        if tkey(x) is None or tkey(x) in (S, D) or K is None or K.kind != E.K_CHALLENGE:
            return None
        # S must be dead after the pair: its next access is a write (or none)
        r = reads_after.get(S, [])
        nr = bisect.bisect_right(r, i + 1)
        nw = last_write.get(S, [])
        nwi = bisect.bisect_right(nw, i + 1)
        next_read = r[nr] if nr < len(r) else n + 1
        next_write = nw[nwi] if nwi < len(nw) else n + 2
        if next_read < next_write:
            return None
        return (x, K, op2, D)

    # runs of consecutive pairs on the same D
    runs, i = [], 0
    while i < n:
        p = is_pair(i)
        if p is None:
            i += 1
            continue
        run = [(i, p)]
        j = i + 2
        while True:
            q = is_pair(j)
            if q is None or q[3] != p[3]:
                break
            run.append((j, q)); j += 2
        runs.append(run)
        i = j
    ntmp3 = prog.ntmp3
    S2 = E.tmp3(ntmp3); ntmp3 += 1
    inserts, removed, replace_first, zero_acc = {}, set(), {}, []
    for run in runs:
        start = run[0][0]
        moved = []
        for pos, (x, K, op2, D) in run:
            j = last_def_before(tkey(x), pos)
            if j is None or start - j < min_distance:
                continue
            moved.append((pos, j, x, K, op2))
        if len(moved) < min_terms:
            continue
        A = E.tmp3(ntmp3); ntmp3 += 1
        zero_acc.append(A)
        for pos, j, x, K, op2 in moved:
            inserts.setdefault(j, []).append((E.MUL, S2, x, K))
            inserts[j].append((op2, A, A, S2))
            removed.add(pos); removed.add(pos + 1)
        D = run[0][1][3]
        replace_first[moved[0][0]] = (E.ADD, E.tmp3(D[1]), E.tmp3(D[1]), A)
    if not zero_acc:
        return prog
    out = E.Program()
    for A in zero_acc:                                          # A = 0 (challenge 0 minus itself: an Fp^3 zero without a constant pool)
        out.sub(A, E.challenge(0), E.challenge(0))
    for i, (op, d, a, b) in enumerate(ins):
        if i in replace_first:
            out.emit(*replace_first[i])
        if i not in removed:
            out.emit(op, d, a, b)
        for t in inserts.get(i, ()):
            out.emit(*t)
    out.ntmp1 = max(out.ntmp1, prog.ntmp1)
    out.ntmp3 = max(out.ntmp3, ntmp3)
    return out


# ---------------------------------------------------------------------------------------------------------------------------
def segment_traffic(prog_or_code, seg=256):
    """offline model of the ahead-of-time form's HBM traffic per row: (column words loaded, summed over segments of `seg` instructions;
    temporary words live across segment boundaries).  Each segment kernel loads a staged column word once and keeps it in a register."""
    code = prog_or_code.code if hasattr(prog_or_code, "code") else prog_or_code
    ins = list(E.decode(code))
    cols, n = 0, len(ins)
    width = {E.K_TMP1: 1, E.K_TMP3: 3}
    last_use, first_def = {}, {}
    for i, (op, d, a, b) in enumerate(ins):
        for o in (a, b):
            if o is not None and o.kind in width:
                last_use[(o.kind, o.params[0], first_def.get((o.kind, o.params[0]), -1))] = i
        if d.kind in width:
            first_def[(d.kind, d.params[0])] = i
    for s0 in range(0, n, seg):
        cs = set()
        for op, d, a, b in ins[s0:s0 + seg]:
            for o in (a, b):
                if o is not None and o.kind in _COL:
                    for c in range(o.dim):
                        cs.add((o.kind in (E.K_POLS, E.K_POLS_PRIME), o.kind in (E.K_CONST2NS, E.K_CONST2NS_PRIME), tuple(o.params), c))
        cols += len(cs)
    tmps = 0
    for (kind, slot, dpos), lu in last_use.items():
        if dpos >= 0 and lu // seg != dpos // seg:
            tmps += width[kind] * (lu // seg - dpos // seg >= 1)
    return cols, tmps
