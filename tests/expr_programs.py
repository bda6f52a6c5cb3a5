"""Seeded random constraint-expression programs for the VM parity tests (tests only).

Layout of the synthetic StepsParams: `pols` has `nin` read-only columns (some read as 3-wide extension
groups), then written columns: each instruction that stores to pols gets a fresh column group so that no
row reads what another row writes (the reference's generated programs have the same property: the parser
loops are `omp parallel for` over rows)."""
import numpy as np

from xlayer_prover_amd import expr as E


def random_program(seed, n, ninstr, nin=24, ntmp1=12, ntmp3=6, nconst=5, nconst2=4, nch=4, npub=3, nev=5, nstore=40, prime_dst=True):
    rng = np.random.default_rng(seed)
    width = nin + 3 * nstore
    prog = E.Program()
    live1, live3 = set(), set()
    next_store = [nin]

    def src(dim):
        while True:
            if dim == 1:
                k = rng.integers(0, 11)
                if k == 0 and live1: return E.tmp1(int(rng.choice(sorted(live1))))
                if k == 1: return E.pols(int(rng.integers(0, nin)), width)
                if k == 2: return E.pols_prime(int(rng.integers(0, nin)), int(rng.integers(1, 4)), n, width)
                if k == 3: return E.const(int(rng.integers(0, nconst)))
                if k == 4: return E.const_prime(int(rng.integers(0, nconst)), 1, n)
                if k == 5: return E.const2ns(int(rng.integers(0, nconst2)))
                if k == 6: return E.const2ns_prime(int(rng.integers(0, nconst2)), 2, n)
                if k == 7: return E.number(int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2)))   # incl. >= p
                if k == 8: return E.public(int(rng.integers(0, npub)))
                if k == 9: return E.x_n() if rng.integers(0, 2) else E.x_2ns()
                if k == 10: return E.zi()
            else:
                k = rng.integers(0, 8)
                if k == 0 and live3: return E.tmp3(int(rng.choice(sorted(live3))))
                if k == 1: return E.pols(int(rng.integers(0, nin - 2)), width, 3)
                if k == 2: return E.pols_prime(int(rng.integers(0, nin - 2)), 1, n, width, 3)
                if k == 3: return E.challenge(int(rng.integers(0, nch)))
                if k == 4: return E.evaluation(int(rng.integers(0, nev)))
                if k == 5: return E.x_div_x_sub_xi()
                if k == 6: return E.x_div_x_sub_wxi()
                if k == 7 and live3: return E.tmp3(int(rng.choice(sorted(live3))))

    for it in range(ninstr):
        op = int(rng.integers(0, 4))
        da = 1 if rng.integers(0, 2) else 3
        db = 1 if rng.integers(0, 2) else 3
        if op == E.COPY:
            db = da
        dd = max(da, db)
        a = src(da)
        b = None if op == E.COPY else src(db)
        r = rng.integers(0, 10)
        if r < 7 or next_store[0] + 3 > width:
            if dd == 1:
                i = int(rng.integers(0, ntmp1)); dst = E.tmp1(i); live1.add(i)
            else:
                i = int(rng.integers(0, ntmp3)); dst = E.tmp3(i); live3.add(i)
        elif r == 7:
            dst = E.pols(next_store[0], width, dd); next_store[0] += 3
        elif r == 8 and prime_dst:
            dst = E.pols_prime(next_store[0], int(rng.integers(1, 3)), n, width, dd); next_store[0] += 3
        else:
            if dd == 3:
                dst = E.q_2ns() if rng.integers(0, 2) else E.f_2ns()
            else:
                dst = E.pols(next_store[0], width, 1); next_store[0] += 3
        prog.emit(op, dst, a, b)
    return prog, width


def make_buffers(orc, seed, n, width, nconst=5, nconst2=4, nch=4, npub=3, nev=5, zi_size=4):
    """name -> uint64 ndarray (inputs are arbitrary 64-bit words: non-canonical values included)"""
    def f(cnt, k):
        return orc.fill(cnt, seed + k)
    raw = np.random.default_rng(seed).integers(0, 2**64, size=n * width, dtype=np.uint64)   # any 64-bit word
    return {
        "pols": raw, "const_pols": f(n * nconst, 1), "const_pols_2ns": f(n * nconst2, 2), "challenges": f(3 * nch, 3),
        "public_inputs": f(npub, 4), "evals": f(3 * nev, 5), "x_n": f(n, 6), "x_2ns": f(n, 7), "zi": f(zi_size, 8),
        "x_div_x_sub_xi": f(3 * n, 9), "x_div_x_sub_wxi": f(3 * n, 10), "q_2ns": f(3 * n, 11), "f_2ns": f(3 * n, 12),
    }


def accumulation_program(seed, n, nterms, nin=24, ntmp1=8, ntmp3=6, nch=4, nev=5, nstore=40):
    """Chains of acc (+|-)= (challenge | evaluation) x base-field value -- what the ahead-of-time generator turns into carry-free
    accumulators (csrc/expr.hip, "lazy accumulation") -- with everything that makes that hard: values in the base field and in Fp^3 (the accumulators themselves included), an accumulator hopping between temporaries,
    an old state read after the accumulator moved on, copies of a state, products read twice or never added, sums of two products,
    accumulators stored or multiplied mid-chain, several chains interleaved (more than the generator keeps alive), base-field bases."""
    rng = np.random.default_rng(seed)
    width = nin + 3 * nstore
    prog = E.Program()
    next_store = [nin]

    def coef():
        return E.challenge(int(rng.integers(0, nch))) if rng.integers(0, 3) else E.evaluation(int(rng.integers(0, nev)))

    def value():
        k = rng.integers(0, 5)
        if k == 0: return E.pols(int(rng.integers(0, nin)), width)
        if k == 1: return E.pols_prime(int(rng.integers(0, nin)), 1, n, width)
        if k == 2: return E.number(int(rng.integers(0, 2**63)) * 2 + 1)
        if k == 3: return E.x_n()
        return E.tmp1(int(rng.integers(0, ntmp1)))
    for i in range(ntmp1):
        prog.emit(E.MUL, E.tmp1(i), E.pols(int(rng.integers(0, nin)), width), E.pols(int(rng.integers(0, nin)), width))
    for i in range(ntmp3):
        prog.emit(E.COPY, E.tmp3(i), E.pols(int(rng.integers(0, nin - 2)), width, 3))
    for it in range(nterms):
        acc = int(rng.integers(0, ntmp3 - 1))
        prodt = ntmp3 - 1                                   # the temporary the products go through
        val = value() if rng.integers(0, 3) else (E.pols(int(rng.integers(0, nin - 2)), width, 3) if rng.integers(0, 2) else E.tmp3(int(rng.integers(0, ntmp3 - 1))))
        prog.emit(E.MUL, E.tmp3(prodt), coef(), val) if rng.integers(0, 2) else prog.emit(E.MUL, E.tmp3(prodt), val, coef())
        r = rng.integers(0, 20)
        if r < 11:                                          # in place
            prog.emit(E.ADD, E.tmp3(acc), E.tmp3(acc), E.tmp3(prodt)) if rng.integers(0, 2) else prog.emit(E.ADD, E.tmp3(acc), E.tmp3(prodt), E.tmp3(acc))
        elif r < 13:
            prog.emit(E.SUB, E.tmp3(acc), E.tmp3(acc), E.tmp3(prodt))
        elif r == 13:                                       # product - accumulator: stays plain
            prog.emit(E.SUB, E.tmp3(acc), E.tmp3(prodt), E.tmp3(acc))
        elif r == 14:                                       # hop: the old temporary keeps the old state and is read later
            other = (acc + 1) % (ntmp3 - 1)
            prog.emit(E.ADD, E.tmp3(other), E.tmp3(acc), E.tmp3(prodt))
        elif r == 15:                                       # the product is read again, and two products are added
            prog.emit(E.ADD, E.tmp3(acc), E.tmp3(prodt), E.tmp3(prodt))
            prog.emit(E.MUL, E.tmp3(acc), E.tmp3(acc), E.tmp3(prodt))
        elif r == 16:                                       # a copy of a state, then the accumulator moves on
            other = (acc + 2) % (ntmp3 - 1)
            prog.emit(E.COPY, E.tmp3(other), E.tmp3(acc))
            prog.emit(E.ADD, E.tmp3(acc), E.tmp3(acc), E.tmp3(prodt))
        elif r == 17:                                       # base-field base
            prog.emit(E.ADD, E.tmp3(acc), value(), E.tmp3(prodt))
        elif r == 18 and next_store[0] + 3 <= width:        # a state is stored mid-chain
            prog.emit(E.ADD, E.tmp3(acc), E.tmp3(acc), E.tmp3(prodt))
            prog.emit(E.COPY, E.pols(next_store[0], width, 3), E.tmp3(acc)); next_store[0] += 3
        else:                                               # a product nobody adds
            if next_store[0] + 3 <= width:
                prog.emit(E.COPY, E.pols(next_store[0], width, 3), E.tmp3(prodt)); next_store[0] += 3
    for i in range(ntmp3 - 1):
        if next_store[0] + 3 <= width:
            prog.emit(E.COPY, E.pols(next_store[0], width, 3), E.tmp3(i)); next_store[0] += 3
    prog.emit(E.COPY, E.q_2ns(), E.tmp3(0))
    prog.emit(E.COPY, E.f_2ns(), E.tmp3(1))
    return prog, width


def long_chain_program(seed, n, nterms, nin=24, nstore=4):
    """ONE accumulator taking nterms products in place: more than the 1 000 the generator gives a carry-free accumulator before it folds it
    and starts a new one with the folded words as its base"""
    rng = np.random.default_rng(seed)
    width = nin + 3 * nstore
    prog = E.Program()
    prog.emit(E.COPY, E.tmp3(0), E.pols(0, width, 3))
    for it in range(nterms):
        prog.emit(E.MUL, E.tmp3(1), E.challenge(int(rng.integers(0, 4))), E.pols(int(rng.integers(0, nin)), width))
        prog.emit(E.SUB if it % 7 == 3 else E.ADD, E.tmp3(0), E.tmp3(0), E.tmp3(1))
    prog.emit(E.COPY, E.q_2ns(), E.tmp3(0))
    return prog, width
