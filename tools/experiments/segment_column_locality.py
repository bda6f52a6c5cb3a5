import sys, collections
sys.path.insert(0, "/root/repo")
from tools import zkevm_geometry as G
from xlayer_prover_amd import expr as E
from xlayer_prover_amd import expr_opt as O
geo, _ = G.load(23, True, "zkevm")
opt = G.load_opt(23, True, "zkevm")
COL = {E.K_POLS, E.K_POLS_PRIME, E.K_CONST, E.K_CONST_PRIME, E.K_CONST2NS, E.K_CONST2NS_PRIME}
def analyse(name, code, SEG=256):
    ins = list(E.decode(code))
    n = len(ins)
    tot_words = tot_sect8 = tot_sect16 = 0
    distinct = set()
    wr = set()
    per_seg = []
    for s0 in range(0, n, SEG):
        cs = set()
        for op, d, a, b in ins[s0:s0+SEG]:
            for o in (a, b):
                if o is not None and o.kind in COL:
                    p = o.params
                    if o.kind in (E.K_POLS, E.K_POLS_PRIME):
                        key = ("pols", p[-1] if o.kind == E.K_POLS_PRIME else p[1]); off = p[0]; nx = p[1] if o.kind == E.K_POLS_PRIME else 0
                    else:
                        key = ("c2" if o.kind in (E.K_CONST2NS, E.K_CONST2NS_PRIME) else "c1", 0); off = p[0]; nx = p[1] if len(p) > 1 else 0
                    for c in range(o.dim):
                        cs.add((key, off + c, nx))
            if d.kind in COL:
                p = d.params
                for c in range(d.dim): wr.add((d.kind, p[0] + c))
        distinct |= {(k, o) for k, o, nx in cs}
        words = len(cs)
        s8 = len({(k, o // 8, nx) for k, o, nx in cs})
        s16 = len({(k, o // 16, nx) for k, o, nx in cs})
        tot_words += words; tot_sect8 += s8; tot_sect16 += s16
        per_seg.append((words, s8))
    print(f"{name}: {n} instr, {len(per_seg)} segs, col words loaded {tot_words}, distinct {len(distinct)}, written {len(wr)}, "
          f"64B sectors x8 = {tot_sect8*8} (f={tot_sect8*8/tot_words:.2f}), 128B lines x16 = {tot_sect16*16} (f={tot_sect16*16/tot_words:.2f}); traffic model {O.segment_traffic(code, SEG)}")
    return per_seg
for st in ["step2prev", "step3prev", "step3", "step42ns"]:
    code = opt[st][0][0] if st in opt else geo[st][0]
    for SEG in (256, 1024):
        analyse(st + f"_opt seg{SEG}", opt[st][0][0], SEG)
