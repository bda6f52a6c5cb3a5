"""Program-level optimisation for the constraint-expression VM: Horner chains -> dot products.

The FRI-polynomial program of a STARK (step52ns; starks.cpp:374-380, generated code in
src/starkpil/zkevm/chelpers/zkevm.chelpers.step52ns.parser.cpp) is a few long Horner chains in a challenge v,

    acc = (...((Y0 v + Y1) v + Y2) v + ...) ,   Y_k = col_k  or  col_k - evals[e_k],

i.e. one Fp^3 x Fp^3 multiplication per committed column and row (2 990 of them for the zkEVM, ~75 % of the program).
Since v and the evaluations do not depend on the row, the same value is

    acc = sum_k col_k * v^(K-k)  -  D ,      D = sum_k evals[e_k] * v^(K-k) ,

with the powers of v and D computed ONCE per proof: a base-field column then costs 3 base multiplications instead of a
9-multiplication (Karatsuba 6) extension product.  Field arithmetic is exact, so the rewritten program produces the same
canonical values as the original.

`horner_to_dot(code, ntmp1, ntmp3, nch)` returns
    row program   -- reads the table as CHALLENGE operands with indices >= nch,
    table program -- row independent; run it on ONE row with `pols` = the extended challenge buffer
                     (3 * n_ext words, the first 3 * nch hold the proof's challenges); it fills entries nch .. n_ext-1,
    n_ext.
Both are ordinary VM programs: nothing changes in the interpreter or the code generator."""
from . import expr as E

_COL = (E.K_POLS, E.K_POLS_PRIME, E.K_CONST, E.K_CONST_PRIME, E.K_CONST2NS, E.K_CONST2NS_PRIME)


class _Lin:
    """value of a tmp3 slot kept symbolic: base * v^bpw + sum X * v^pw - sum evals[e] * v^pw   (v = challenge c)"""
    __slots__ = ("c", "terms", "evals", "bpw")

    def __init__(self, c=None, terms=(), evals=(), bpw=None):
        self.c, self.terms, self.evals, self.bpw = c, list(terms), list(evals), bpw

    def shifted(self, c):
        return _Lin(c, [(x, p + 1) for x, p in self.terms], [(e, p + 1) for e, p in self.evals], None if self.bpw is None else self.bpw + 1)

    def plus(self, o):
        return _Lin(self.c if self.c is not None else o.c, self.terms + o.terms, self.evals + o.evals, self.bpw)


def horner_to_dot(code, ntmp1, ntmp3, nch):
    out, tab = E.Program(), E.Program()
    scratch = E.tmp3(ntmp3)                                  # one more extension temporary for the products
    state = {}                                               # tmp3 slot -> _Lin (not yet materialised)
    powers = {}                                              # (c, j) -> table index of v_c^j
    n_ext = [nch]
    stats = {"chains": 0, "ext_mul_removed": 0}

    def entry():
        n_ext[0] += 1
        return n_ext[0] - 1

    def tab_pols(idx): return E.pols(3 * idx, 3, 3)

    def power(c, j):
        """CHALLENGE operand holding v_c^j (j >= 1); table entries are created in increasing j"""
        if j == 1:
            return E.challenge(c)
        if (c, j) not in powers:
            prev = E.challenge(c) if j == 2 else (power(c, j - 1), tab_pols(powers[(c, j - 1)]))[1]
            idx = entry()
            tab.mul(tab_pols(idx), prev, E.challenge(c))
            powers[(c, j)] = idx
        return E.challenge(powers[(c, j)])

    def tab_power(c, j):
        power(c, j)
        return E.challenge(c) if j == 1 else tab_pols(powers[(c, j)])

    def materialise(slot):
        lin = state.pop(slot, None)
        if lin is None:
            return
        dst = E.tmp3(slot)
        hi_t = [(x, p) for x, p in lin.terms if p >= 1]
        lo_t = [x for x, p in lin.terms if p == 0]
        hi_e = [(e, p) for e, p in lin.evals if p >= 1]
        lo_e = [e for e, p in lin.evals if p == 0]
        for p in sorted({p for _, p in hi_t} | {p for _, p in hi_e} | ({lin.bpw} if lin.bpw else set())):
            for j in range(2, p + 1):
                power(lin.c, j)                              # fill the table in order
        started = False
        if lin.bpw is not None:
            if lin.bpw >= 1:
                out.mul(dst, dst, power(lin.c, lin.bpw))
            started = True
        for x, p in hi_t:
            if not started:
                out.mul(dst, x, power(lin.c, p)); started = True
            else:
                out.mul(scratch, x, power(lin.c, p)).add(dst, dst, scratch)
        if not started:                                      # only row-0-power terms: the original `col - eval` form
            assert lo_t and lo_e, "unexpected symbolic value"
            out.sub(dst, lo_t.pop(0), E.evaluation(lo_e.pop(0)))
        for x in lo_t:
            out.add(dst, dst, x)
        for e in lo_e:
            out.sub(dst, dst, E.evaluation(e))
        if hi_e:                                             # D = sum evals[e] * v^p, one table entry
            t0, t1 = E.tmp3(0), E.tmp3(1)
            for k, (e, p) in enumerate(hi_e):
                if k == 0:
                    tab.mul(t0, E.evaluation(e), tab_power(lin.c, p))
                else:
                    tab.mul(t1, E.evaluation(e), tab_power(lin.c, p)).add(t0, t0, t1)
            idx = entry()
            tab.copy(tab_pols(idx), t0)
            out.sub(dst, dst, E.challenge(idx))
        if len(hi_t) > 1:
            stats["chains"] += 1

    def is_t3(o, slot=None): return o is not None and o.kind == E.K_TMP3 and (slot is None or o.params[0] == slot)
    def is_col(o): return o is not None and o.kind in _COL

    for op, d, a, b in E.decode(code):
        if is_t3(d):
            s = d.params[0]
            # T = col - evals[e]
            if op == E.SUB and is_col(a) and b.kind == E.K_EVAL:
                state[s] = _Lin(None, [(a, 0)], [(b.params[0], 0)])
                continue
            if op == E.MUL and b is not None and b.kind == E.K_CHALLENGE and b.params[0] < nch:
                c = b.params[0]
                if is_col(a):                                # A = col * v
                    state[s] = _Lin(c, [(a, 1)])
                    continue
                if is_t3(a):
                    src = a.params[0]
                    lin = state.get(src)
                    if lin is not None and lin.c in (None, c):
                        state[s] = lin.shifted(c); stats["ext_mul_removed"] += 1
                        continue
                    if src == s:                             # materialised accumulator: keep it as the base of a new chain
                        materialise(s)
                        state[s] = _Lin(c, bpw=1)
                        continue
            if op == E.ADD and is_t3(a, s) and s in state:
                lin = state[s]
                if is_col(b):
                    state[s] = lin.plus(_Lin(None, [(b, 0)]))
                    continue
                if is_t3(b) and b.params[0] in state and b.params[0] != s:
                    o = state[b.params[0]]
                    if o.bpw is None and (o.c is None or lin.c is None or o.c == lin.c):
                        state[s] = lin.plus(o)
                        continue
        # anything else: its tmp3 sources must hold real values; its destination becomes a real value
        for o in (a, b):
            if is_t3(o):
                materialise(o.params[0])
        if is_t3(d):
            state.pop(d.params[0], None)
        out.emit(op, d, a, b)
    out.ntmp1 = max(out.ntmp1, ntmp1)
    out.ntmp3 = max(out.ntmp3, ntmp3 + 1)
    tab.ntmp3 = max(tab.ntmp3, 2)
    return out, tab, n_ext[0], stats
