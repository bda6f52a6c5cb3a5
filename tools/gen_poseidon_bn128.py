"""Poseidon-BN128 constants (circomlib's optimized form, t = 2..17) -> xlayer_prover_amd/chelpers/poseidon_bn128.bin.

Reads the tables C / S / M / P of the reference's src/poseidon_opt/constants_opt.hpp (RawFr::Element initialisers: four 64-bit
limbs, little-endian limb order, Montgomery form) and writes them as data:
    "XLPBN128" | u64 version = 1 | for t = 2..17: u64 nC, nS, nM, nP | elements (4 x u64 LE each) in the order t = 2: C S M P, t = 3: ...
M and P keep the reference's index order m[j][i] (poseidon_opt.cpp:89-99).  Product data like the converted step programs: generated
where the reference tree is mounted, git-ignored, travels to the GPU box with the tree."""
import re
import struct
import sys


def parse_tables(text):
    pos = {}
    for name in ("C", "M", "P", "S"):
        m = re.search(r"std::vector<[^=;]*>\s+%s\s*=\s*\{" % name, text)
        if not m:
            raise SystemExit(f"table {name} not found")
        pos[name] = m.end() - 1
    order = sorted(pos, key=lambda k: pos[k])
    tables = {}
    elem = re.compile(r"\{\s*(0x[0-9A-Fa-f]+)\s*,\s*(0x[0-9A-Fa-f]+)\s*,\s*(0x[0-9A-Fa-f]+)\s*,\s*(0x[0-9A-Fa-f]+)\s*\}")
    for k, name in enumerate(order):
        start = pos[name]
        end = pos[order[k + 1]] if k + 1 < len(order) else len(text)
        body = text[start:end]
        # split per t: depth-1 groups of the outer initializer
        groups, depth, g0 = [], 0, None
        i = 0
        while i < len(body):
            ch = body[i]
            if ch == "{":
                depth += 1
                if depth == 2:
                    g0 = i
            elif ch == "}":
                if depth == 2:
                    groups.append(body[g0:i + 1])
                depth -= 1
                if depth == 0:
                    break
            i += 1
        tables[name] = [[tuple(int(x, 16) for x in m.groups()) for m in elem.finditer(g)] for g in groups]
    return tables


def main():
    src, out = sys.argv[1], sys.argv[2]
    t = parse_tables(open(src).read())
    n_rounds_p = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]   # poseidon_opt.hpp:18
    blob = bytearray(b"XLPBN128") + struct.pack("<Q", 1)
    body = bytearray()
    for k in range(16):
        tt = k + 2
        C, S, M, P = t["C"][k], t["S"][k], t["M"][k], t["P"][k]
        assert len(C) == 8 * tt + n_rounds_p[k], (tt, len(C))
        assert len(S) == (2 * tt - 1) * n_rounds_p[k], (tt, len(S))
        assert len(M) == tt * tt and len(P) == tt * tt, (tt, len(M), len(P))
        blob += struct.pack("<4Q", len(C), len(S), len(M), len(P))
        for tab in (C, S, M, P):
            for e in tab:
                body += struct.pack("<4Q", *e)
    open(out, "wb").write(bytes(blob) + bytes(body))
    print(f"{out}: {len(body) // 32} elements")


if __name__ == "__main__":
    main()
