#!/usr/bin/env python3
"""Everything host/starks.hpp needs to run one STARK of genBatchProof, in one directory ("bundle"): the counterpart of the
reference's config/<stark>/ directory (starkinfo.json + chelpers), for the geometries of tools/zkevm_geometry.py:
    <out>/<circ>.starkinfo.json          stark_info.cpp schema (synthesized: tools/zkevm_geometry.py synth_starkinfo)
    <out>/<circ>.<step>.xlpexpr          the converted reference program relocated to this slab
    <out>/<circ>.<step>.opt.xlpexpr, .tab.xlpexpr        its dot-product form (xlayer_prover_amd/expr_opt.py)
    <out>/<circ>.step52ns.dots, .drow.xlpexpr, .dtab.xlpexpr   the FRI polynomial as table program + xlp_dev_dot_columns + row program
    <out>/<circ>.steps.json              which of these XlpSteps runs per step, and the ahead-of-time plugin of each
Needs the converted programs (xlayer_prover_amd/chelpers/*.xlpexpr); no GPU.
    python tools/make_stark_bundle.py --circuit zkevm --log-n 23 --aliased"""
import argparse
import json
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import zkevm_geometry as G
from xlayer_prover_amd import expr as E
from xlayer_prover_amd import programs as R

MAGIC = int.from_bytes(b"XLPEXPR1", "little")
MAGIC_TAB = int.from_bytes(b"XLPTAB01", "little")
MAGIC_DOTS = int.from_bytes(b"XLPDOTS1", "little")


def write_prog(path, code, n1, n3):
    code = np.ascontiguousarray(code, dtype="<u8")
    with open(path, "wb") as f:
        f.write(struct.pack("<4Q", MAGIC, n1, n3, code.size)); f.write(code.tobytes())


def write_tab(path, prog, n_ext):
    code = np.ascontiguousarray(prog.code, dtype="<u8")
    with open(path, "wb") as f:
        f.write(struct.pack("<5Q", MAGIC_TAB, prog.ntmp1, prog.ntmp3, code.size, n_ext)); f.write(code.tobytes())


def geo_tag(log_n, aliased=False, world=0, colmajor=False):
    return f"geo{log_n}{'a' if aliased else ''}{f'w{world}' if world else ''}{'c' if colmajor else ''}"


def bundle_dir(circ, log_n, aliased, world=0, colmajor=False):
    return os.path.join(R.REF_DIR, "bundles", f"{circ}.{geo_tag(log_n, aliased, world, colmajor)}")


def make(circ, log_n, aliased, out=None, fri_steps=None, queries=None, use_opt=True, use_dots=True, world=0, colmajor=True, plugin_dir=None):
    """world > 0: the bundle of ONE RANK of a proof row-sharded over `world` ranks (every rank uses the same files): window starkinfo,
    programs relocated to the window slab (tools/zkevm_geometry.py layout_window), and per step the next-row stores whose halo rows go
    back to their owner.
    colmajor: COLUMN-MAJOR device sections (include/xlp_gl.h, XLP_EXPR_COLMAJOR): the same slab packing, every section stored as columns
    (the extended ones coset-major); the programs are converted last (tools/zkevm_geometry.py to_colmajor), the starkinfo and the manifest
    say "colMajor": true.  With world > 0: the window slab as columns -- a column of a base-domain section = the rank's rows + the halo row,
    of an extended one = 2^ext coset blocks of (the rank's rows + the halo row)"""
    if circ == "recursivef":
        colmajor = False                                       # its sections feed Poseidon-BN128 trees row by row (StarkRecursiveF): the reference's layout
    out = out or bundle_dir(circ, log_n, aliased, world, colmajor)
    os.makedirs(out, exist_ok=True)
    geo = geo_tag(log_n, aliased, world, colmajor)
    LW = G.layout_window(log_n, world, circ) if world else None
    progs, info = G.load(log_n, aliased, circ, LW)
    si = G.synth_starkinfo(circ, log_n, aliased, fri_steps, queries, world)
    L = LW or G.layout(log_n, aliased, circ)
    ext = G.CIRCUITS[circ]["ext"]
    cm = (lambda code: G.to_colmajor(code, L, ext)) if colmajor else (lambda code: code)
    if colmajor:
        si["xlp"]["colMajor"] = True
    json.dump(si, open(os.path.join(out, f"{circ}.starkinfo.json"), "w"))
    man = {}
    rel = lambda p: os.path.relpath(p, out)

    def plugin(name):
        so = os.path.join(plugin_dir or R.REF_DIR, name)
        return rel(so) if os.path.exists(so) else None
    for st, (code, n1, n3) in progs.items():
        write_prog(os.path.join(out, f"{circ}.{st}.xlpexpr"), cm(code), n1, n3)
        man[st] = {"prog": f"{circ}.{st}.xlpexpr", "plugin": plugin(f"{circ}.{st}.{geo}.so")}
    if use_opt:
        for st, ((rc, r1, r3), tabp, n_ext, stats) in G.load_opt(log_n, aliased, circ, LW).items():
            write_prog(os.path.join(out, f"{circ}.{st}.opt.xlpexpr"), cm(rc), r1, r3)
            e = {"prog": f"{circ}.{st}.opt.xlpexpr", "plugin": plugin(f"{circ}.{st}_opt.{geo}.so"), "n_ext": n_ext}
            if tabp.ninstr:
                write_tab(os.path.join(out, f"{circ}.{st}.tab.xlpexpr"), tabp, n_ext)
                e["tab"] = f"{circ}.{st}.tab.xlpexpr"
            man[st] = e
    if use_opt and use_dots:
        tabp, dots, rowp, n_ext = G.load_fri_dots(log_n, aliased, circ, LW)
        write_tab(os.path.join(out, f"{circ}.step52ns.dtab.xlpexpr"), tabp, n_ext)
        rows_ext = L["cm1_2ns"][2]                              # (a window: the rank's extended rows + halo)
        drow = rowp.code
        if colmajor:                                           # the row program reads the kernel's output: 3 * ngroups columns of rows_ext words
            drow = G.to_colmajor(drow, {"dots": (0, 3 * dots["ngroups"], rows_ext)}, ext)
        write_prog(os.path.join(out, f"{circ}.step52ns.drow.xlpexpr"), drow, rowp.ntmp1, rowp.ntmp3)
        ents = []
        for x, g, k in dots["entries"]:
            if x.kind == E.K_POLS:
                ents.append((0, x.params[0], x.params[1], x.dim, g, k))
            else:
                assert x.kind == E.K_CONST2NS
                ents.append((1, x.params[0], 0, 1, g, k))
        if colmajor:
            ents = G.dots_to_colmajor(ents, L)
        ents = [v for e in ents for v in e]
        with open(os.path.join(out, f"{circ}.step52ns.dots"), "wb") as f:
            f.write(struct.pack("<3Q", MAGIC_DOTS, dots["ngroups"], len(dots["entries"]))); f.write(np.array(ents, dtype="<u8").tobytes())
        man["step52ns"] = {"prog": f"{circ}.step52ns.drow.xlpexpr", "tab": f"{circ}.step52ns.dtab.xlpexpr", "n_ext": n_ext, "dots": f"{circ}.step52ns.dots",
                           "tab_plugin": plugin(f"{circ}.step52ns_tab.{geo}.so"), "plugin": plugin(f"{circ}.step52ns_drow.{geo}.so")}
    if world:
        for st, (code, n1, n3) in progs.items():               # (the rewritten forms store to the same places)
            man[st]["nextRowStores"] = [list(x) for x in G.next_row_stores(code, LW)]
    json.dump({"steps": man, "colMajor": bool(colmajor)}, open(os.path.join(out, f"{circ}.steps.json"), "w"), indent=1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--circuit", default="zkevm")
    ap.add_argument("--log-n", type=int, default=0)
    ap.add_argument("--aliased", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--no-opt", action="store_true")
    ap.add_argument("--no-dots", action="store_true")
    ap.add_argument("--world", type=int, default=0, help="the per-rank bundle of a proof row-sharded over this many ranks (window layout)")
    ap.add_argument("--colmajor", action="store_true", help="column-major device sections")
    a = ap.parse_args()
    ln = a.log_n or G.CIRCUITS[a.circuit]["log_n"]
    print(make(a.circuit, ln, a.aliased, a.out or None, use_opt=not a.no_opt, use_dots=not a.no_dots, world=a.world, colmajor=a.colmajor))


if __name__ == "__main__":
    main()
