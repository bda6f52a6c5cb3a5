"""Assembler for the constraint-expression VM (encoding: include/xlp_gl.h, "constraint-expression VM").

Host-side helper only: it builds the uint64 program that xlp_expr_create uploads; evaluation happens in
csrc/expr.hip.  Used by tools/chelpers_to_xlp.py (conversion of the reference's generated op/arg tables,
src/starkpil/zkevm/chelpers/*.parser.hpp) and by the tests.
"""
import numpy as np

COPY, ADD, SUB, MUL = 0, 1, 2, 3
(K_NONE, K_TMP1, K_TMP3, K_POLS, K_POLS_PRIME, K_CONST, K_CONST_PRIME, K_CONST2NS, K_CONST2NS_PRIME, K_NUMBER, K_CHALLENGE, K_PUBLIC,
 K_EVAL, K_X_N, K_X_2NS, K_ZI, K_XDIVXSUBXI, K_XDIVXSUBWXI, K_Q_2NS, K_F_2NS) = range(20)
KIND_PARAMS = [0, 1, 1, 2, 4, 1, 3, 1, 3, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
FIXED_DIM = {K_TMP1: 1, K_TMP3: 3, K_CONST: 1, K_CONST_PRIME: 1, K_CONST2NS: 1, K_CONST2NS_PRIME: 1, K_NUMBER: 1, K_CHALLENGE: 3, K_PUBLIC: 1,
             K_EVAL: 3, K_X_N: 1, K_X_2NS: 1, K_ZI: 1, K_XDIVXSUBXI: 3, K_XDIVXSUBWXI: 3, K_Q_2NS: 3, K_F_2NS: 3}


class Operand:
    __slots__ = ("kind", "dim", "params")

    def __init__(self, kind, params=(), dim=None):
        if len(params) != KIND_PARAMS[kind]:
            raise ValueError(f"operand kind {kind} takes {KIND_PARAMS[kind]} parameters, got {len(params)}")
        self.kind = kind
        self.dim = FIXED_DIM.get(kind, dim)
        if self.dim not in (1, 3):
            raise ValueError("operand dimension must be 1 or 3")
        self.params = [int(p) for p in params]


def tmp1(i): return Operand(K_TMP1, (i,))
def tmp3(i): return Operand(K_TMP3, (i,))
def pols(offset, stride, dim=1): return Operand(K_POLS, (offset, stride), dim)
def pols_prime(offset, nxt, n, stride, dim=1): return Operand(K_POLS_PRIME, (offset, nxt, n, stride), dim)
def const(col): return Operand(K_CONST, (col,))
def const_prime(col, nxt, n): return Operand(K_CONST_PRIME, (col, nxt, n))
def const2ns(col): return Operand(K_CONST2NS, (col,))
def const2ns_prime(col, nxt, n): return Operand(K_CONST2NS_PRIME, (col, nxt, n))
def number(v): return Operand(K_NUMBER, (v,))
def challenge(k): return Operand(K_CHALLENGE, (k,))
def public(k): return Operand(K_PUBLIC, (k,))
def evaluation(k): return Operand(K_EVAL, (k,))
def x_n(): return Operand(K_X_N)
def x_2ns(): return Operand(K_X_2NS)
def zi(): return Operand(K_ZI)
def x_div_x_sub_xi(): return Operand(K_XDIVXSUBXI)
def x_div_x_sub_wxi(): return Operand(K_XDIVXSUBWXI)
def q_2ns(): return Operand(K_Q_2NS)
def f_2ns(): return Operand(K_F_2NS)


class Program:
    """Accumulates instructions; `.code` is the uint64 array, `.ntmp1/.ntmp3` the temporaries it needs."""

    def __init__(self):
        self.words = []
        self.ntmp1 = 0
        self.ntmp3 = 0
        self.ninstr = 0

    def emit(self, op, dst, a, b=None):
        bk, bd, bp = (K_NONE, 0, []) if b is None else (b.kind, b.dim, b.params)
        head = op | dst.kind << 8 | a.kind << 16 | bk << 24 | dst.dim << 32 | a.dim << 34 | bd << 36
        self.words += [head] + dst.params + a.params + bp
        for o in (dst, a) + (() if b is None else (b,)):
            if o.kind == K_TMP1:
                self.ntmp1 = max(self.ntmp1, o.params[0] + 1)
            if o.kind == K_TMP3:
                self.ntmp3 = max(self.ntmp3, o.params[0] + 1)
        self.ninstr += 1
        return self

    def copy(self, dst, a): return self.emit(COPY, dst, a)
    def add(self, dst, a, b): return self.emit(ADD, dst, a, b)
    def sub(self, dst, a, b): return self.emit(SUB, dst, a, b)
    def mul(self, dst, a, b): return self.emit(MUL, dst, a, b)

    @property
    def code(self):
        return np.array(self.words, dtype=np.uint64)


def decode(code):
    """uint64 program -> iterator of (op, dst, a, b) with Operand objects (b is None for COPY)"""
    pc, n = 0, len(code)
    while pc < n:
        h = int(code[pc])
        q = pc + 1
        ops = []
        for s in range(3):
            kind, dim = (h >> (8 + 8 * s)) & 255, (h >> (32 + 2 * s)) & 3
            cnt = KIND_PARAMS[kind]
            ops.append(None if kind == K_NONE else Operand(kind, [int(v) for v in code[q:q + cnt]], dim))
            q += cnt
        yield h & 255, ops[0], ops[1], ops[2]
        pc = q
