"""Run a converted reference program (tools/chelpers_to_xlp.py output) on a small synthetic trace.

The checked-in zkEVM tables address a 2^23-row slab (section bases and `% N` moduli are baked into the
arguments); `shrink` relocates every pols section to a packed slab of `n` rows and rewrites the prime moduli,
keeping columns, strides, next-row distances and everything else as they are.  Tests / benchmarks only."""
import os

import numpy as np  # noqa: F401

from xlayer_prover_amd import expr as E

from xlayer_prover_amd.programs import MAGIC, REF_DIR, operands, read_program  # noqa: F401  (re-exported for the tests)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shrink_many(codes, n):
    """one relocation for several programs over the same slab -> ([new codes], info)"""
    codes = [c.copy() for c in codes]
    allops = [list(operands(c)) for c in codes]
    by_stride = {}
    for code, ops in zip(codes, allops):
        for _, _, kind, dim, q in ops:
            if kind == E.K_POLS:
                by_stride.setdefault(int(code[q + 1]), set()).add(int(code[q]))
            elif kind == E.K_POLS_PRIME:
                by_stride.setdefault(int(code[q + 3]), set()).add(int(code[q]))
    remap, top = {}, 0
    for stride, offs in sorted(by_stride.items()):
        lo = None
        for o in sorted(offs):
            if lo is None or o - lo >= stride:
                lo, base = o, top
                top += n * stride
            remap[(stride, o)] = base + (o - lo)
    info = {"pols_size": top + 4, "n_const": 1, "n_const2": 1, "nch": 1, "npub": 1, "nev": 1, "max_next": 0}
    for code, ops in zip(codes, allops):
        for _, _, kind, dim, q in ops:
            if kind == E.K_POLS:
                code[q] = remap[(int(code[q + 1]), int(code[q]))]
            elif kind == E.K_POLS_PRIME:
                code[q] = remap[(int(code[q + 3]), int(code[q]))]
                info["max_next"] = max(info["max_next"], int(code[q + 1]))
                code[q + 2] = n
            elif kind in (E.K_CONST, E.K_CONST_PRIME):
                info["n_const"] = max(info["n_const"], int(code[q]) + 1)
            elif kind in (E.K_CONST2NS, E.K_CONST2NS_PRIME):
                info["n_const2"] = max(info["n_const2"], int(code[q]) + 1)
            elif kind == E.K_CHALLENGE:
                info["nch"] = max(info["nch"], int(code[q]) + 1)
            elif kind == E.K_PUBLIC:
                info["npub"] = max(info["npub"], int(code[q]) + 1)
            elif kind == E.K_EVAL:
                info["nev"] = max(info["nev"], int(code[q]) + 1)
            if kind in (E.K_CONST_PRIME, E.K_CONST2NS_PRIME):
                info["max_next"] = max(info["max_next"], int(code[q + 1]))
                code[q + 2] = n
    assert info["max_next"] < n
    return codes, info


def shrink(code, n):
    """-> (new code, info) with info = sizes the buffers need"""
    codes, info = shrink_many([code], n)
    return codes[0], info


def buffers(fill, seed, n, info):
    """fill(count, seed) -> uint64 ndarray; returns name -> ndarray for xlp_steps_params / the oracle"""
    return {
        "pols": fill(info["pols_size"], seed), "const_pols": fill(n * info["n_const"], seed + 1), "const_pols_2ns": fill(n * info["n_const2"], seed + 2),
        "challenges": fill(3 * info["nch"], seed + 3), "public_inputs": fill(info["npub"], seed + 4), "evals": fill(3 * info["nev"], seed + 5),
        "x_n": fill(n, seed + 6), "x_2ns": fill(n, seed + 7), "zi": fill(2, seed + 8), "x_div_x_sub_xi": fill(3 * n, seed + 9),
        "x_div_x_sub_wxi": fill(3 * n, seed + 10), "q_2ns": fill(3 * n, seed + 11), "f_2ns": fill(3 * n, seed + 12),
    }
