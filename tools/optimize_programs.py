#!/usr/bin/env python3
"""Setup-time pass over converted step programs: Horner chains -> dot products (xlayer_prover_amd/expr_opt.py).
For every <dir>/<name>.<step>.xlpexpr writes, when the rewrite removes extension products,
    <dir>/<name>.<step>.opt.xlpexpr   the row program (same header), reading powers / constants as CHALLENGE operands >= 8
    <dir>/<name>.<step>.tab.xlpexpr   the one-row table program, header "XLPTAB01", ntmp1, ntmp3, ncode, n_ext
`XlpSteps` (xlayer_prover_amd/host/xlp_host.hpp) picks the pair up when it is there.
    python tools/optimize_programs.py <dir> [--nch 8]"""
import argparse
import glob
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xlayer_prover_amd import expr_opt

MAGIC = int.from_bytes(b"XLPEXPR1", "little")
MAGIC_TAB = int.from_bytes(b"XLPTAB01", "little")


def optimise_file(path, nch=8):
    raw = open(path, "rb").read()
    magic, n1, n3, ncode = struct.unpack("<4Q", raw[:32])
    assert magic == MAGIC and len(raw) == 32 + 8 * ncode, path
    code = np.frombuffer(raw, dtype="<u8", offset=32).astype(np.uint64)
    row, tab, n_ext, st = expr_opt.horner_to_dot(code, n1, n3, nch)
    base = path[:-len(".xlpexpr")]
    if st["ext_mul_after"] >= st["ext_mul_before"]:
        return st, False
    with open(base + ".opt.xlpexpr", "wb") as f:
        f.write(struct.pack("<4Q", MAGIC, row.ntmp1, row.ntmp3, row.code.size)); f.write(row.code.astype("<u8").tobytes())
    with open(base + ".tab.xlpexpr", "wb") as f:
        f.write(struct.pack("<5Q", MAGIC_TAB, tab.ntmp1, tab.ntmp3, tab.code.size, n_ext)); f.write(tab.code.astype("<u8").tobytes())
    return st, True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--nch", type=int, default=8)
    a = ap.parse_args()
    for path in sorted(glob.glob(os.path.join(a.dir, "*.xlpexpr"))):
        if path.endswith(".opt.xlpexpr") or path.endswith(".tab.xlpexpr"):
            continue
        st, wrote = optimise_file(path, a.nch)
        print(f"{os.path.basename(path)}: extension products {st['ext_mul_before']} -> {st['ext_mul_after']}" + ("" if wrote else "  (kept as is)"))


if __name__ == "__main__":
    main()
