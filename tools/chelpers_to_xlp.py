#!/usr/bin/env python3
"""Convert the reference's generated constraint-evaluation tables to xlp expression-VM programs.

Input (read where it lies, nothing is copied into the repo):
    <ref>/src/starkpil/<pil>/chelpers/<name>.chelpers.<step>.parser.cpp   the interpreter: `switch (op[kk])` whose
                                                                          cases are single library calls
    <ref>/src/starkpil/<pil>/chelpers/<name>.chelpers.<step>.parser.hpp   NOPS_, NARGS_, NTEMP1_, NTEMP3_, op[], args[]
Output:
    <out>/<name>.<step>.xlpexpr   little-endian uint64: magic, ntmp1, ntmp3, ncode, code[ncode]
                                  (encoding: include/xlp_gl.h, "constraint-expression VM")

How: the generated `_parser_first_avx` function is regular -- every case is one call
`Goldilocks[3]::<op><dims>_avx(dst, src..., strides/offsets...)` (or a few of them on fixed accumulators in
step52ns) followed by `i_args += k`.  Each case is parsed ONCE into a template of VM instructions whose
operand parameters refer to `args[i_args + j]`; the op table is then replayed through the templates.
A case the grammar below does not recognise aborts the conversion (nothing is guessed), and the totals are
checked against NOPS_ / NARGS_.

    python tools/chelpers_to_xlp.py --ref /root/reference --out xlayer_prover_amd/chelpers [--pil zkevm --name zkevm]
"""
import argparse
import os
import re
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xlayer_prover_amd import expr as E  # noqa: E402

MAGIC = int.from_bytes(b"XLPEXPR1", "little")
STEPS = ["step2prev", "step3prev", "step3", "step42ns", "step52ns"]


class Arg:
    """parameter = args[i_args + k]"""
    def __init__(self, k): self.k = k


class Lit:
    def __init__(self, v): self.v = v


class OperandT:
    def __init__(self, kind, params, dim=None, needs_offsets=None):
        self.kind, self.params, self.dim, self.needs_offsets = kind, params, dim, needs_offsets


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([": depth += 1
        if ch in ")]": depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


A = r"A(\d+)"


def classify(tok):
    """-> OperandT, or ('skip',) for stride/offset-count tokens, or ('offsets', n), or None"""
    t = tok.replace(" ", "")
    m = re.fullmatch(r"tmp1\[\(?" + A + r"\)?\]", t)
    if m: return OperandT(E.K_TMP1, [Arg(int(m.group(1)))], 1)
    m = re.fullmatch(r"tmp3\[\(?" + A + r"\)?\]", t)
    if m: return OperandT(E.K_TMP3, [Arg(int(m.group(1)))], 3)
    m = re.fullmatch(r"&params\.pols\[" + A + r"\+i\*" + A + r"\]", t)
    if m: return OperandT(E.K_POLS, [Arg(int(m.group(1))), Arg(int(m.group(2)))])
    if t == "&params.pols[0]": return OperandT(E.K_POLS_PRIME, None, needs_offsets="pols")
    m = re.fullmatch(r"&params\.pConstPols(2ns)?->getElement\((" + A + r"|\d+),i\)", t)
    if m:
        col = Arg(int(m.group(3))) if m.group(3) is not None else Lit(int(m.group(2)))
        return OperandT(E.K_CONST2NS if m.group(1) else E.K_CONST, [col], 1)
    m = re.fullmatch(r"&params\.pConstPols(2ns)?->getElement\(0,0\)", t)
    if m: return OperandT(E.K_CONST2NS_PRIME if m.group(1) else E.K_CONST_PRIME, None, 1, needs_offsets="const")
    m = re.fullmatch(r"Goldilocks::fromU64\(" + A + r"\)", t)
    if m: return OperandT(E.K_NUMBER, [Arg(int(m.group(1)))], 1)
    m = re.fullmatch(r"(?:\(Goldilocks3::Element&\)\*)?params\.challenges\[(" + A + r"|\d+)\]", t)
    if m:
        return OperandT(E.K_CHALLENGE, [Arg(int(m.group(2))) if m.group(2) is not None else Lit(int(m.group(1)))], 3)
    m = re.fullmatch(r"params\.publicInputs\[" + A + r"\]", t)
    if m: return OperandT(E.K_PUBLIC, [Arg(int(m.group(1)))], 1)
    if t == "params.x_n[i]": return OperandT(E.K_X_N, [], 1)
    if t == "params.x_2ns[i]": return OperandT(E.K_X_2NS, [], 1)
    if t == "params.xDivXSubXi[i]": return OperandT(E.K_XDIVXSUBXI, [], 3)
    if t == "params.xDivXSubWXi[i]": return OperandT(E.K_XDIVXSUBWXI, [], 3)
    m = re.fullmatch(r"&evals_\[" + A + r"\*3\]", t)
    if m: return OperandT(E.K_EVAL, [Arg(int(m.group(1)))], 3)
    if t == "evals_": return OperandT(E.K_EVAL, [Lit(0)], 3)
    if t == "&(params.f_2ns[i*3])": return OperandT(E.K_F_2NS, [], 3)
    m = re.fullmatch(r"offsets(\d)", t)
    if m: return ("offsets", int(m.group(1)))
    if re.fullmatch(A, t) or t in ("numpols", "params.pConstPols2ns->numPols()", "params.pConstPols->numPols()", "params.x_n.offset()",
                                   "params.x_2ns.offset()"):
        return ("skip",)
    return None


def fold_registers(toks, acc_base):
    """tmpK_0, tmpK_1, tmpK_2 -> one TMP3 accumulator; challK0_.. (6 tokens) -> CHALLENGE(K)"""
    out, i = [], 0
    while i < len(toks):
        m = re.fullmatch(r"tmp(\d)_0", toks[i])
        if m and toks[i + 1:i + 3] == [f"tmp{m.group(1)}_1", f"tmp{m.group(1)}_2"]:
            out.append(OperandT(E.K_TMP3, [Lit(("acc", int(m.group(1))))], 3)); i += 3; continue
        m = re.fullmatch(r"chall(\d)0_", toks[i])
        if m:
            k = m.group(1)
            if toks[i:i + 6] != [f"chall{k}0_", f"chall{k}1_", f"chall{k}2_", f"chall{k}o0_", f"chall{k}o1_", f"chall{k}o2_"]:
                raise ValueError("unexpected challenge register list: " + ", ".join(toks[i:i + 6]))
            out.append(OperandT(E.K_CHALLENGE, [Lit(int(k))], 3)); i += 6; continue
        out.append(toks[i]); i += 1
    return out


def parse_offsets(expr):
    e = expr.replace(" ", "")
    m = re.fullmatch(A + r"\+\(\(\(i\+j\)\+" + A + r"\)%" + A + r"\)\*" + A, e)
    if m: return ("pols_prime", [Arg(int(g)) for g in m.groups()])
    m = re.fullmatch(A + r"\+\(\(\(i\+j\)\+" + A + r"\)%" + A + r"\)\*numpols", e)
    if m: return ("const_prime", [Arg(int(g)) for g in m.groups()])
    m = re.fullmatch(A + r"\+\(i\+j\)\*" + A, e)
    if m: return ("pols_row", [Arg(int(g)) for g in m.groups()])
    m = re.fullmatch(A + r"\+\(i\+j\)\*numpols", e)
    if m: return ("const_row", [Arg(int(m.group(1)))])
    raise ValueError("unrecognised offsets expression: " + expr)


FUNC = re.compile(r"(Goldilocks3?)::(add|sub|mul|mult|copy)((?:[13]c?){2})?_avx$")


def func_dims(ns, name):
    m = FUNC.fullmatch(f"{ns}::{name}")
    if not m: raise ValueError(f"unrecognised function {ns}::{name}")
    op = {"add": E.ADD, "sub": E.SUB, "mul": E.MUL, "mult": E.MUL, "copy": E.COPY}[m.group(2)]
    if m.group(3):
        d = re.findall(r"[13]", m.group(3))
        return op, int(d[0]), int(d[1])
    return (op, 3, 3) if ns == "Goldilocks3" else (op, 1, 1)


def parse_case(body, argname):
    """-> (list of (op, dst, a, b) templates, total i_args increment).  Fused cases ("// 0, 50": several calls with an
    `i_args += k` between them) are parsed segment by segment with the running argument base."""
    body = re.sub(r"//.*", "", body)
    body = re.sub(argname + r"\[i_args\s*\+\s*(\d+)\]", r"A\1", body)
    body = re.sub(argname + r"\[i_args\]", "A0", body)
    pieces = re.split(r"i_args \+= (\d+);", body)
    if re.search(r"Goldilocks3?::\w+\(", pieces[-1]):
        if len(pieces) > 1: raise ValueError("call after the last i_args increment")
        pieces = [pieces[0], "0", ""]            # case that consumes no argument (fixed accumulators of step52ns)
    templates, base = [], 0
    for seg, inc in zip(pieces[0::2], pieces[1::2]):
        for t in parse_segment(seg):
            for o in t[1:]:
                if o is not None:
                    for prm in o.params:
                        if isinstance(prm, Arg): prm.k += base
            templates.append(t)
        base += int(inc)
    if not templates: raise ValueError("no call found")
    return templates, base


def parse_segment(body):
    if "params.q_2ns" in body:
        # q_2ns[i] = zi.zhInv(i) * tmp3[A0]      (Goldilocks3::mul(ext &, base, ext &), step42ns case 69)
        if not re.search(r"params\.zi\.zhInv\(\(i \+ j\)\)", body) or "tmp3[A0][0]" not in body.replace(" ", "").replace("(A0)", "A0"):
            raise ValueError("unrecognised q_2ns case")
        return [(E.MUL, OperandT(E.K_Q_2NS, [], 3), OperandT(E.K_ZI, [], 1), OperandT(E.K_TMP3, [Arg(0)], 3))]
    offsets = {}
    for m in re.finditer(r"offsets(\d)\[j\]\s*=\s*(.*?);", body):
        offsets[int(m.group(1))] = m.group(2)      # parsed when a call uses it (some are dead stores)
    templates = []
    for m in re.finditer(r"(Goldilocks3?)::(\w+)\((.*)\);", body):
        ns, name, args = m.group(1), m.group(2), m.group(3)
        if name in ("fromU64",): continue
        op, da, db = func_dims(ns, name)
        toks = fold_registers(split_args(args), None)
        operands, pending = [], []
        for t in toks:
            c = t if isinstance(t, OperandT) else classify(t)
            if c is None: raise ValueError(f"unrecognised argument '{t}' in {ns}::{name}({args})")
            if isinstance(c, OperandT):
                operands.append(c)
                if c.needs_offsets: pending.append(c)
            elif c[0] == "offsets":
                if not pending: raise ValueError(f"offsets{c[1]} without an operand in {args}")
                o = pending.pop(0)
                kind, prm = parse_offsets(offsets[c[1]])
                if o.needs_offsets == "pols":
                    if kind == "pols_prime": o.kind, o.params = E.K_POLS_PRIME, prm
                    elif kind == "pols_row": o.kind, o.params = E.K_POLS, prm
                    else: raise ValueError("const offsets on a pols operand")
                else:
                    if kind == "const_row": o.kind = E.K_CONST2NS if o.kind == E.K_CONST2NS_PRIME else E.K_CONST
                    elif kind != "const_prime": raise ValueError("pols offsets on a const operand")
                    o.params = prm
        if pending: raise ValueError("operand without offsets in " + args)
        nsrc = 1 if op == E.COPY else 2
        if len(operands) != 1 + nsrc: raise ValueError(f"{ns}::{name}: expected {1 + nsrc} operands, got {len(operands)} in ({args})")
        dst, a = operands[0], operands[1]
        b = operands[2] if nsrc == 2 else None
        if op == E.COPY:
            da = db = dst.dim or a.dim or (3 if ns == "Goldilocks3" else 1)
        for o, d in ((a, da), (b, db)):
            if o is None: continue
            if o.dim is None: o.dim = d
            elif o.dim != d: raise ValueError(f"{ns}::{name}: operand dimension {o.dim} != {d} in ({args})")
        dd = max(da, db)
        if dst.dim is None: dst.dim = dd
        elif dst.dim != dd: raise ValueError(f"{ns}::{name}: destination dimension {dst.dim} != {dd}")
        templates.append((op, dst, a, b))
    return templates


def read_tables(hpp):
    txt = open(hpp).read()
    d = {k: int(v) for k, v in re.findall(r"#define (NOPS_|NARGS_|NTEMP1_|NTEMP3_) (\d+)", txt)}
    arrays = re.findall(r"uint64_t (\w+)\[(NOPS_|NARGS_)\]\s*=\s*\{(.*?)\};", txt, re.S)
    ops = args = None
    for name, size, data in arrays:
        vals = [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", data)]
        if size == "NOPS_": ops, opname = vals, name
        else: args, argname = vals, name
    assert ops is not None and args is not None, "op/args tables not found in " + hpp
    assert len(ops) == d["NOPS_"] and len(args) == d["NARGS_"], (len(ops), len(args), d)
    return d, ops, args, opname, argname


def convert(cpp, hpp, step):
    d, ops, args, opname, argname = read_tables(hpp)
    src = open(cpp).read()
    m = re.search(r"void \w+::" + step + r"_parser_first_avx\(.*?\)\s*\{(.*?)\n\}\n", src, re.S)
    assert m, "parser function not found in " + cpp
    parts = re.split(r"\n\s*case (\d+):\s*\n", m.group(1))[1:]
    cases = {}
    for num, body in zip(parts[0::2], parts[1::2]):
        try:
            cases[int(num)] = parse_case(body, argname)
        except ValueError as e:
            raise SystemExit(f"{os.path.basename(cpp)} case {num}: {e}")
    ntmp1, ntmp3 = d.get("NTEMP1_", 0), d.get("NTEMP3_", 0)
    acc = {}     # fixed accumulators tmpK_* of step52ns -> extra TMP3 slots after the table's own
    prog = E.Program()
    ia = 0

    def inst(o):
        if o is None: return None
        prm = []
        for p in o.params:
            if isinstance(p, Arg): prm.append(args[ia + p.k])
            elif isinstance(p.v, tuple): prm.append(ntmp3 + acc.setdefault(p.v[1], len(acc)))
            else: prm.append(p.v)
        return E.Operand(o.kind, prm, o.dim)

    for kk, opc in enumerate(ops):
        if opc not in cases: raise SystemExit(f"{os.path.basename(hpp)}: op {opc} at {kk} has no case")
        templates, inc = cases[opc]
        for op, dst, a, b in templates:
            prog.emit(op, inst(dst), inst(a), inst(b))
        ia += inc
    if ia != d["NARGS_"]: raise SystemExit(f"{os.path.basename(hpp)}: consumed {ia} args, table has {d['NARGS_']}")
    ntmp1, ntmp3 = max(ntmp1, prog.ntmp1), max(ntmp3 + len(acc), prog.ntmp3)
    return prog, ntmp1, ntmp3, {"nops": d["NOPS_"], "nargs": d["NARGS_"], "cases": len(cases), "ninstr": prog.ninstr, "ncode": len(prog.words),
                                "ntmp1": ntmp1, "ntmp3": ntmp3}


def write_program(path, code, ntmp1, ntmp3):
    with open(path, "wb") as f:
        f.write(struct.pack("<4Q", MAGIC, ntmp1, ntmp3, len(code)))
        f.write(code.astype("<u8").tobytes())


def read_program(path):
    import numpy as np
    raw = open(path, "rb").read()
    magic, ntmp1, ntmp3, ncode = struct.unpack("<4Q", raw[:32])
    if magic != MAGIC or len(raw) != 32 + 8 * ncode: raise ValueError(path + ": not an xlp expression program")
    return np.frombuffer(raw, dtype="<u8", offset=32).astype(np.uint64), ntmp1, ntmp3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--pil", default="zkevm", help="directory under src/starkpil")
    ap.add_argument("--name", default="zkevm", help="file prefix of <name>.chelpers.<step>.parser.*")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    base = os.path.join(args.ref, "src", "starkpil", args.pil, "chelpers")
    os.makedirs(args.out, exist_ok=True)
    for step in STEPS:
        cpp = os.path.join(base, f"{args.name}.chelpers.{step}.parser.cpp")
        hpp = os.path.join(base, f"{args.name}.chelpers.{step}.parser.hpp")
        if not (os.path.exists(cpp) and os.path.exists(hpp)):
            print("skip", step, "(no parser tables)"); continue
        prog, ntmp1, ntmp3, info = convert(cpp, hpp, step)
        dst = os.path.join(args.out, f"{args.name}.{step}.xlpexpr")
        write_program(dst, prog.code, ntmp1, ntmp3)
        print(step, info, "->", dst)


if __name__ == "__main__":
    main()
