#!/usr/bin/env python3
"""Generates xlayer_prover_amd/csrc/dft_asm.inc: the in-register Goldilocks radix-8 / radix-16
DFT networks as hand-scheduled gfx950 inline assembly.

Why: a Goldilocks add/sub is a 32-bit carry chain.  On gfx940+ a VALU instruction that reads an
SGPR pair (carry-in, v_cndmask mask) written by a VALU instruction needs 2 wait states in between
(LLVM GCNHazardRecognizer: VALUWriteSGPRVALURead).  hipcc keeps every chain in VCC (VOP2 encodings)
and pads with `s_nop 1`, so a canonical add costs ~14 issue slots instead of 6 (measured:
profiles/r1/ubench_field.txt).  Here every macro-op (butterfly, multiply by a power of two) is
a list of instructions with explicit SGPR-pair carries (VOP3 encodings); G >= 3 independent macro-ops
are interleaved round-robin so the two wait states are filled with useful work.

Arithmetic (canonical in, canonical out; p = 2^64 - 2^32 + 1, EPS = 2^32 - 1 = 2^64 mod p):
  bfly   : (a, b) -> (a + b, a - b)
  shiftA : x * 2^k, 0 < k < 32      = (y1:y0) + y2*EPS                      y = x << k  (96 bits)
  shiftB : x * 2^k, 32 <= k < 64    = (y0:0) + y1*EPS - y2                  y = x << (k-32)
  shiftC : x * 2^k, 64 <= k < 96    = y0*EPS - (y2:y1)                      y = x << (k-64)
(2^96 = -1 mod p.)  The DFT network is the DIF network of ntt_common.hpp::dft_regs: on return
x[bitrev(k)] = X[k].
"""
import sys

MIN_GAP = 3   # positions between a VALU SGPR write and a VALU read of it (2 wait states)


class Block:
    """one inline-asm statement: virtual registers -> operand numbers"""

    def __init__(self):
        self.lists = []      # instruction lists of the macro-ops, each item = (text, sdefs, suses)
        self.io = []         # names of in/out u32 variables, in operand order
        self.vtmp = 0
        self.stmp = 0

    def io_reg(self, name):
        if name not in self.io:
            self.io.append(name)
        return "IO:" + name

    def vt(self):
        self.vtmp += 1
        return f"VT:{self.vtmp - 1}"

    def st(self):
        self.stmp += 1
        return f"ST:{self.stmp - 1}"


def ins(lst, text, sdef=(), suse=(), salu=False):
    lst.append((text, tuple(sdef), tuple(suse), salu))


def bfly(B, a, b):
    """a = (alo, ahi), b = (blo, bhi) io names; in place: a <- a+b, b <- a-b"""
    L = []
    al, ah, bl, bh = (B.io_reg(a + "l"), B.io_reg(a + "h"), B.io_reg(b + "l"), B.io_reg(b + "h"))
    s0, s1, t0, t1, m = B.vt(), B.vt(), B.vt(), B.vt(), B.vt()
    cS, cD, cT, cE = B.st(), B.st(), B.st(), B.st()
    ins(L, f"v_add_co_u32 {s0}, {cS}, {al}, {bl}", [cS])
    ins(L, f"v_sub_co_u32 {bl}, {cD}, {al}, {bl}", [cD])                  # d0 overwrites bl (al, bl read above)
    ins(L, f"v_addc_co_u32 {s1}, {cS}, {ah}, {bh}, {cS}", [cS], [cS])
    ins(L, f"v_subb_co_u32 {bh}, {cD}, {ah}, {bh}, {cD}", [cD], [cD])    # d1 overwrites bh
    ins(L, f"v_add_co_u32 {t0}, {cT}, -1, {s0}", [cT])                   # t = s + EPS (= s - p mod 2^64)
    ins(L, f"v_cndmask_b32 {m}, 0, -1, {cD}", [], [cD])                  # m = borrow ? EPS : 0
    ins(L, f"v_addc_co_u32 {t1}, {cT}, 0, {s1}, {cT}", [cT], [cT])
    ins(L, f"v_sub_co_u32 {bl}, {cE}, {bl}, {m}", [cE])
    ins(L, f"s_or_b64 {cS}, {cS}, {cT}", [], [], salu=True)              # SALU: no wait states needed
    ins(L, f"v_subbrev_co_u32 {bh}, {cE}, 0, {bh}, {cE}", [cE], [cE])
    ins(L, f"v_cndmask_b32 {al}, {s0}, {t0}, {cS}", [], [])
    ins(L, f"v_cndmask_b32 {ah}, {s1}, {t1}, {cS}", [], [])
    # the s_or result feeds the two cndmasks: SALU write -> VALU read needs no wait state
    B.lists.append(L)


def shl96(B, L, xl, xh, r):
    """(y2:y1:y0) = x << r, 0 <= r < 32; returns names (may alias inputs when r == 0)"""
    if r == 0:
        return xl, xh, None
    y0, y1, y2 = B.vt(), B.vt(), B.vt()
    ins(L, f"v_lshlrev_b32 {y0}, {r}, {xl}")
    ins(L, f"v_alignbit_b32 {y1}, {xh}, {xl}, {32 - r}")
    ins(L, f"v_lshrrev_b32 {y2}, {32 - r}, {xh}")
    return y0, y1, y2


def canon_select(B, L, rl, rh, carry, outl, outh):
    """out = (carry or r >= p) ? r + EPS : r      (r < 2p as a 65-bit value with `carry`)"""
    t0, t1 = B.vt(), B.vt()
    cT = B.st()
    ins(L, f"v_add_co_u32 {t0}, {cT}, -1, {rl}", [cT])
    ins(L, f"v_addc_co_u32 {t1}, {cT}, 0, {rh}, {cT}", [cT], [cT])
    if carry is not None:
        ins(L, f"s_or_b64 {cT}, {cT}, {carry}", [], [], salu=True)
    ins(L, f"v_cndmask_b32 {outl}, {rl}, {t0}, {cT}", [], [] if carry is not None else [cT])
    ins(L, f"v_cndmask_b32 {outh}, {rh}, {t1}, {cT}", [], [] if carry is not None else [cT])


def shift(B, x, k):
    """x <- x * 2^k mod p (canonical in/out), 0 < k < 96"""
    L = []
    xl, xh = B.io_reg(x + "l"), B.io_reg(x + "h")
    q, r = divmod(k, 32)
    y0, y1, y2 = shl96(B, L, xl, xh, r)
    if q == 0:
        # (y1:y0) + y2*EPS ; W = y2*EPS = (y2 - [y2 != 0] : -y2)
        assert y2 is not None
        wl, wh, rl, rh = B.vt(), B.vt(), B.vt(), B.vt()
        cW, cR = B.st(), B.st()
        ins(L, f"v_sub_co_u32 {wl}, {cW}, 0, {y2}", [cW])
        ins(L, f"v_subbrev_co_u32 {wh}, {cW}, 0, {y2}, {cW}", [cW], [cW])
        ins(L, f"v_add_co_u32 {rl}, {cR}, {y0}, {wl}", [cR])
        ins(L, f"v_addc_co_u32 {rh}, {cR}, {y1}, {wh}, {cR}", [cR], [cR])
        canon_select(B, L, rl, rh, cR, xl, xh)
    elif q == 2:
        # y0*EPS - (y2:y1) ; T = y0*EPS canonical, W = (y2:y1) < 2^52 canonical  -> canonical subtract
        tl, th, m = B.vt(), B.vt(), B.vt()
        cT, cD, cE = B.st(), B.st(), B.st()
        w_hi = y2 if y2 is not None else "0"
        ins(L, f"v_sub_co_u32 {tl}, {cT}, 0, {y0}", [cT])
        ins(L, f"v_subbrev_co_u32 {th}, {cT}, 0, {y0}, {cT}", [cT], [cT])
        ins(L, f"v_sub_co_u32 {tl}, {cD}, {tl}, {y1}", [cD])
        if y2 is not None:
            ins(L, f"v_subb_co_u32 {th}, {cD}, {th}, {y2}, {cD}", [cD], [cD])
        else:
            ins(L, f"v_subbrev_co_u32 {th}, {cD}, 0, {th}, {cD}", [cD], [cD])
        ins(L, f"v_cndmask_b32 {m}, 0, -1, {cD}", [], [cD])
        ins(L, f"v_sub_co_u32 {xl}, {cE}, {tl}, {m}", [cE])
        ins(L, f"v_subbrev_co_u32 {xh}, {cE}, 0, {th}, {cE}", [cE], [cE])
    else:
        # V = (y0:0) + y1*EPS - y2  in (-2^28, 2p).   T = y1*EPS ; S = T + (y0:0) (carry C) ;
        # R = S - y2 (borrow Bw) ; K = C - Bw :  K=-1 -> R - EPS ; K=+1 -> R + EPS ; K=0 -> canon(R)
        tl, th, sh, rl, rh = B.vt(), B.vt(), B.vt(), B.vt(), B.vt()
        pl, ph, ul, uh = B.vt(), B.vt(), B.vt(), B.vt()
        cT, cC, cB, cP, cU, cX = B.st(), B.st(), B.st(), B.st(), B.st(), B.st()
        ins(L, f"v_sub_co_u32 {tl}, {cT}, 0, {y1}", [cT])
        ins(L, f"v_subbrev_co_u32 {th}, {cT}, 0, {y1}, {cT}", [cT], [cT])
        ins(L, f"v_add_co_u32 {sh}, {cC}, {th}, {y0}", [cC])                 # S = (sh : tl), carry C
        if y2 is not None:
            ins(L, f"v_sub_co_u32 {rl}, {cB}, {tl}, {y2}", [cB])
            ins(L, f"v_subbrev_co_u32 {rh}, {cB}, 0, {sh}, {cB}", [cB], [cB])   # borrow Bw
        else:
            rl, rh = tl, sh
        ins(L, f"v_add_co_u32 {pl}, {cP}, -1, {rl}", [cP])                   # P = R + EPS, carry cP
        ins(L, f"v_addc_co_u32 {ph}, {cP}, 0, {rh}, {cP}", [cP], [cP])
        if y2 is not None:
            ins(L, f"v_sub_co_u32 {ul}, {cU}, {rl}, -1", [cU])                   # U = R - EPS
            ins(L, f"v_subbrev_co_u32 {uh}, {cU}, 0, {rh}, {cU}", [cU], [cU])
            # selU = ~C & Bw ; selP = (C & ~Bw) | (~(C ^ Bw) & cP)
            ins(L, f"s_andn2_b64 {cU}, {cB}, {cC}", [], [], salu=True)           # selU
            ins(L, f"s_xor_b64 {cX}, {cC}, {cB}", [], [], salu=True)             # K != 0
            ins(L, f"s_andn2_b64 {cP}, {cP}, {cX}", [], [], salu=True)           # K == 0 & cP
            ins(L, f"s_andn2_b64 {cX}, {cC}, {cB}", [], [], salu=True)           # C & ~Bw
            ins(L, f"s_or_b64 {cP}, {cP}, {cX}", [], [], salu=True)              # selP
            ins(L, f"v_cndmask_b32 {rl}, {rl}, {pl}, {cP}")
            ins(L, f"v_cndmask_b32 {rh}, {rh}, {ph}, {cP}")
            ins(L, f"v_cndmask_b32 {xl}, {rl}, {ul}, {cU}")
            ins(L, f"v_cndmask_b32 {xh}, {rh}, {uh}, {cU}")
        else:
            ins(L, f"s_or_b64 {cP}, {cP}, {cC}", [], [], salu=True)
            ins(L, f"v_cndmask_b32 {xl}, {rl}, {pl}, {cP}")
            ins(L, f"v_cndmask_b32 {xh}, {rh}, {ph}, {cP}")
    B.lists.append(L)


def schedule(B):
    """round-robin interleave of the macro-op lists; s_nop where a VALU-written SGPR would be read too early"""
    out = []
    last_def = {}   # sgpr -> position of the VALU instruction that wrote it
    idx = [0] * len(B.lists)
    live = list(range(len(B.lists)))
    cur = 0
    while live:
        # pick the next list (round robin) whose head has no hazard; else any; else nop
        order = live[cur % len(live):] + live[:cur % len(live)]
        pick = None
        for li in order:
            text, sdef, suse, salu = B.lists[li][idx[li]]
            pos = len(out)
            ok = salu or all(pos - last_def.get(s, -100) >= MIN_GAP for s in suse)
            if ok:
                pick = li
                break
        if pick is None:
            out.append("s_nop 0")
            continue
        text, sdef, suse, salu = B.lists[pick][idx[pick]]
        pos = len(out)
        out.append(text)
        if not salu:
            for s in sdef:
                last_def[s] = pos
        else:
            for s in sdef:
                last_def.pop(s, None)
        idx[pick] += 1
        if idx[pick] == len(B.lists[pick]):
            live.remove(pick)
            cur = 0 if not live else cur % len(live)
        else:
            cur = (live.index(pick) + 1) % len(live)
    return out


def emit_block(B, fn_name, params):
    """params: list of u64 parameter names (references), in the order of the C++ signature"""
    lines = schedule(B)
    nio, nvt, nst = len(B.io), B.vtmp, B.stmp
    opnum = {}
    for i, n in enumerate(B.io):
        opnum["IO:" + n] = i
    for i in range(nvt):
        opnum[f"VT:{i}"] = nio + i
    for i in range(nst):
        opnum[f"ST:{i}"] = nio + nvt + i
    def sub(t):
        # replace longest tokens first
        for key in sorted(opnum, key=len, reverse=True):
            t = t.replace(key, f"%{opnum[key]}")
        return t
    body = "\\n\\t".join(sub(t) for t in lines)
    nn = sum(1 for t in lines if t.startswith("s_nop"))
    src = []
    sig = ", ".join(f"u64& {p}" for p in params)
    src.append(f"// {len(lines)} instructions ({nn} s_nop), {nvt} VGPR temps, {nst} SGPR pairs")
    src.append(f"__device__ __forceinline__ void {fn_name}({sig})")
    src.append("{")
    for p in params:
        src.append(f"    u32 {p}l = (u32){p}, {p}h = (u32)({p} >> 32);")
    if nvt:
        src.append("    u32 " + ", ".join(f"vt{i}" for i in range(nvt)) + ";")
    if nst:
        src.append("    u64 " + ", ".join(f"st{i}" for i in range(nst)) + ";")
    outs = [f'"+v"({n})' for n in B.io] + [f'"=&v"(vt{i})' for i in range(nvt)] + [f'"=&s"(st{i})' for i in range(nst)]
    src.append(f'    asm("{body}"')
    src.append("        : " + ", ".join(outs))
    src.append("        :")
    src.append("        : );")
    for p in params:
        src.append(f"    {p} = ((u64){p}h << 32) | {p}l;")
    src.append("}")
    return "\n".join(src) + "\n"


def gen():
    out = ["// GENERATED by tools/gen_dft_asm.py -- do not edit.\n"]
    # ---- butterfly groups of 4 and 2+..: bfly4(a0,b0,a1,b1,a2,b2,a3,b3)
    B = Block()
    names = []
    for i in range(4):
        bfly(B, f"a{i}", f"b{i}")
        names += [f"a{i}", f"b{i}"]
    out.append(emit_block(B, "bfly4_asm", names))
    # ---- shift groups used by the networks (k lists)
    groups = {
        "shift_12_24_36_48": [12, 24, 36, 48],
        "shift_60_72_84": [60, 72, 84],
        "shift_24_48_72": [24, 48, 72],
        "shift_48x4": [48, 48, 48, 48],
        "shift_48x2": [48, 48],
    }
    for fn, ks in groups.items():
        B = Block()
        names = []
        for i, k in enumerate(ks):
            shift(B, f"x{i}", k)
            names.append(f"x{i}")
        out.append(emit_block(B, fn + "_asm", names))
    # ---- the networks
    out.append('''
// DIF networks: x[brev(k)] = X[k] on return (same contract as dft_regs<D>)
__device__ __forceinline__ void dft16_asm(u64 (&x)[16])
{
    // stage 0: pairs (i, i+8), twiddle 2^(12 i) on the difference
    bfly4_asm(x[0], x[8], x[1], x[9], x[2], x[10], x[3], x[11]);
    bfly4_asm(x[4], x[12], x[5], x[13], x[6], x[14], x[7], x[15]);
    shift_12_24_36_48_asm(x[9], x[10], x[11], x[12]);
    shift_60_72_84_asm(x[13], x[14], x[15]);
    // stage 1: pairs (i, i+4) in each half, twiddle 2^(24 i)
    bfly4_asm(x[0], x[4], x[1], x[5], x[2], x[6], x[3], x[7]);
    bfly4_asm(x[8], x[12], x[9], x[13], x[10], x[14], x[11], x[15]);
    shift_24_48_72_asm(x[5], x[6], x[7]);
    shift_24_48_72_asm(x[13], x[14], x[15]);
    // stage 2: pairs (i, i+2) in each quarter, twiddle 2^(48 i)
    bfly4_asm(x[0], x[2], x[1], x[3], x[4], x[6], x[5], x[7]);
    bfly4_asm(x[8], x[10], x[9], x[11], x[12], x[14], x[13], x[15]);
    shift_48x4_asm(x[3], x[7], x[11], x[15]);
    // stage 3: adjacent pairs
    bfly4_asm(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
    bfly4_asm(x[8], x[9], x[10], x[11], x[12], x[13], x[14], x[15]);
}

__device__ __forceinline__ void dft8_asm(u64 (&x)[8])
{
    bfly4_asm(x[0], x[4], x[1], x[5], x[2], x[6], x[3], x[7]);
    shift_24_48_72_asm(x[5], x[6], x[7]);
    bfly4_asm(x[0], x[2], x[1], x[3], x[4], x[6], x[5], x[7]);
    shift_48x2_asm(x[3], x[7]);
    bfly4_asm(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
}
''')
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "xlayer_prover_amd/csrc/dft_asm.inc"
    open(path, "w").write(gen())
    print("wrote", path)
