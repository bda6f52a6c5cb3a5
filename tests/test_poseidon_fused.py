"""csrc/poseidon_fused.inc (tools/gen_poseidon_fused.py): three partial rounds of Poseidon-Goldilocks as one linear layer.  The tables
the HIP permutation reads must reproduce the oracle's round-by-round permutation, and must be what the generator writes today."""
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "xlayer_prover_amd", "csrc", "poseidon_fused.inc")
P = 0xFFFFFFFF00000001


def _tables():
    txt = re.sub(r"/\*.*?\*/", "", open(INC).read(), flags=re.S)
    a, b = txt.split("#define XLP_POSEIDON_FUSED_U64")
    u32 = [int(v) for v in re.findall(r"(\d+)u\b", a)]
    u64 = [int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", b)]
    assert len(u32) == 193 and len(u64) == 98
    return u32, u64


def _consts():
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "xlayer_prover_amd", "csrc", "poseidon_constants.inc")).read(), flags=re.S)
    return [int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", txt)]


def _perm_fused(state):
    """the permutation as csrc/poseidon.hip::permute walks it"""
    u32, u64 = _tables()
    C = _consts()
    circ = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
    M = [[circ[(j - i) % 12] + (8 if i == 0 and j == 0 else 0) for j in range(12)] for i in range(12)]
    mds = lambda s, c: [(sum(M[i][j] * s[j] for j in range(12)) + (c[i] if c else 0)) % P for i in range(12)]
    p7 = lambda x: pow(x, 7, P)
    R1, R2, m00 = u32[:12], u32[12:24], u32[24]
    B3, B2, A1 = u32[25:169], u32[169:181], u32[181:193]
    s = [(x + c) % P for x, c in zip(state, C[:12])]
    for r in range(4):
        s = mds([p7(x) for x in s], C[12 * (r + 1):12 * (r + 2)])
    for g in range(7):
        kc = u64[14 * g:14 * g + 14]
        s[0] = p7(s[0])
        y2 = p7((sum(a * b for a, b in zip(s, R1)) + kc[0]) % P)
        y3 = p7((sum(a * b for a, b in zip(s, R2)) + y2 * m00 + kc[1]) % P)
        s = [(sum(s[j] * B3[12 * i + j] for j in range(12)) + y2 * B2[i] + y3 * A1[i] + kc[2 + i]) % P for i in range(12)]
    s[0] = p7(s[0])
    s = mds(s, C[12 * 26:12 * 27])
    for r in range(26, 29):
        s = mds([p7(x) for x in s], C[12 * (r + 1):12 * (r + 2)])
    return mds([p7(x) for x in s], None)


def test_fused_tables_reproduce_the_permutation(orc):
    rng = np.random.default_rng(11)
    cases = [[0] * 12, [P - 1] * 12, list(range(12))] + [[int(v) % P for v in rng.integers(0, 2**63, 12, dtype=np.uint64) * 2 + 1] for _ in range(5)]
    for st in cases:
        assert _perm_fused(st) == [int(v) for v in orc.perm(st)]


def test_fused_tables_fit_the_multiply_add_chains():
    u32, _ = _tables()
    B3, B2, A1 = u32[25:169], u32[169:181], u32[181:193]
    worst = max(sum(B3[12 * i:12 * i + 12]) + B2[i] + A1[i] for i in range(12))
    assert (2**32 - 1) * worst + 2**32 < 2**57            # L and H of recombine() stay below 2^57


def test_fused_tables_are_what_the_generator_writes(tmp_path):
    out = tmp_path / "f.inc"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_poseidon_fused.py"), str(out)], check=True, capture_output=True)
    assert out.read_text() == open(INC).read()
