"""BN128 side of StarkRecursiveF: Poseidon over the BN254 scalar field, MerkleTreeBN128, TranscriptBN128.

CPU (`not gpu`): the oracle's permutation against circomlib's published known answers (the only external pin that exists:
the reference snapshot holds no BN128 golden data), tree-layout arithmetic against the reference's formulas.
GPU: the HIP kernels against the oracle, through the C ABI, bit-exact on the Montgomery-limb bytes."""
import os

import numpy as np
import pytest

from oracle import bn128_oracle as bo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONST = os.path.join(ROOT, "xlayer_prover_amd", "chelpers", "poseidon_bn128.bin")
needs_constants = pytest.mark.skipif(not os.path.exists(CONST), reason="poseidon_bn128.bin not generated (needs the reference tree at build time)")


@pytest.fixture(scope="module")
def K():
    return bo.Constants(CONST)


@needs_constants
def test_oracle_permutation_known_answers(K):
    """circomlib / circomlibjs poseidon test vectors: poseidon([1, 2]) and poseidon([1, 2, 3, 4]) (state[0] = 0 capacity)"""
    assert bo.poseidon(K, [0, 1, 2])[0] == 0x115cc0f5e7d690413df64c6b9662e9cf2a3617f2743245519e19607a4417189a
    assert bo.poseidon(K, [0, 1, 2, 3, 4])[0] == 0x299c867db6c1fdd79dcefa40e4510b9837e60ebb1ce0663dbaa525df65250465


def test_montgomery_round_trip():
    for x in (0, 1, 2, bo.R_MOD - 1, 0x1234567890ABCDEF << 100):
        assert bo.from_mont_limbs(bo.to_mont_limbs(x)) == x % bo.R_MOD
    assert bo.to_mont_limbs(1) == [0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f]   # R mod r (fr.asm's `one`)


@pytest.mark.parametrize("h,exp", [(1, 1 * 16 + 0), (16, 16 + 1), (256, 256 + 16 + 1), (1 << 19, (1 << 19) + (1 << 15) + (1 << 11) + 128 + 16 + 1)])
def test_num_nodes(h, exp):
    if h == 1:
        assert bo.get_num_nodes(1) == 16        # n = 1: nextN * ARITY, loop not entered
    else:
        assert bo.get_num_nodes(h) == exp


@needs_constants
def test_oracle_tree_and_proof_are_consistent(K):
    """a proof's groups contain the node of the level below at position idx & 15 and hash (t = 17) to the group's parent"""
    rng = np.random.default_rng(5)
    h, w = 64, 7
    src = rng.integers(0, 0xFFFFFFFF00000001, size=(h, w), dtype=np.uint64)
    nodes = bo.merkelize(K, src)
    assert len(nodes) == bo.get_num_nodes(h)
    for idx in (0, 17, 63):
        mp = bo.merkle_proof(nodes, idx, h)
        cur, i = bo.linear_hash_row(K, src[idx]), idx
        for group in mp:
            assert group[i & 15] == cur
            cur = bo.poseidon(K, [0] + list(group))[0]
            i >>= 4
        assert cur == nodes[-1]


@needs_constants
def test_oracle_transcript_is_deterministic_and_order_sensitive(K):
    a, b = bo.TranscriptBN128(K), bo.TranscriptBN128(K)
    a.put([1, 2, 3]); b.put([1, 3, 2])
    fa, fb = a.get_field(), b.get_field()
    assert fa != fb and all(x < (1 << 64) for x in fa)
    c = bo.TranscriptBN128(K)
    c.put([1, 2, 3])
    assert c.get_field() == fa
    perms = c.get_permutations(43, 20)
    assert len(perms) == 43 and all(p < (1 << 20) for p in perms)


# ------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def xlp():
    from xlayer_prover_amd import capi
    capi.init(0)
    return capi


@pytest.mark.gpu
@needs_constants
@pytest.mark.parametrize("t", [2, 3, 5, 9, 16, 17])
def test_gpu_permutation_vs_oracle(xlp, K, t):
    rng = np.random.default_rng(100 + t)
    count = 70                                     # more than one wave
    vals = [[int.from_bytes(rng.bytes(32), "little") % bo.R_MOD for _ in range(t)] for _ in range(count)]
    vals[0] = [0] * t
    vals[1] = [bo.R_MOD - 1] * t
    if t == 3:
        vals[2] = [0, 1, 2]
    st = np.array([[bo.to_mont_limbs(x) for x in row] for row in vals], dtype=np.uint64)
    got = xlp.poseidon_bn128(st, t)
    for i in (0, 1, 2, 3, 63, 64, 69):
        exp = bo.poseidon(K, vals[i])
        assert [bo.from_mont_limbs(got[i, j]) for j in range(t)] == exp
        assert (got[i] == np.array([bo.to_mont_limbs(x) for x in exp], dtype=np.uint64)).all()     # canonical Montgomery limbs
    if t == 3:
        assert bo.from_mont_limbs(got[2, 0]) == 0x115cc0f5e7d690413df64c6b9662e9cf2a3617f2743245519e19607a4417189a


@pytest.mark.gpu
@needs_constants
@pytest.mark.parametrize("h,w", [(16, 12), (256, 3), (256, 4), (64, 5), (32, 48), (16, 49), (32, 51), (512, 18), (1, 12), (2, 9), (8, 100)])
def test_gpu_merkle_bn128_vs_oracle(xlp, K, h, w):
    """every leaf rule (w <= 4 raw, one chunk, exactly 16 Fr, 16 + a t = 2 tail, several chunks) and level shapes (single batch of < 16,
    exact powers of 16, a padded top level)"""
    rng = np.random.default_rng(h * 1000 + w)
    src = rng.integers(0, 0xFFFFFFFF00000001, size=(h, w), dtype=np.uint64)
    src[0, :] = 0xFFFFFFFF00000000
    got = xlp.merkle_bn128(src, w, h)
    exp = bo.nodes_to_limbs(bo.merkelize(K, src))
    assert got.shape == exp.shape
    assert (got == exp).all()


@pytest.mark.gpu
@needs_constants
def test_gpu_merkle_bn128_larger_tree_properties(xlp, K):
    """2^14 x 12 (the recursivef leaf shape): root and sampled paths against the oracle's rules without building the whole oracle tree"""
    h, w = 1 << 14, 12
    rng = np.random.default_rng(77)
    src = rng.integers(0, 0xFFFFFFFF00000001, size=(h, w), dtype=np.uint64)
    got = xlp.merkle_bn128(src, w, h)
    nodes = [bo.from_mont_limbs(r) for r in got]
    for idx in (0, 1, 4097, h - 1):
        assert nodes[idx] == bo.linear_hash_row(K, src[idx])
        cur, i = nodes[idx], idx
        for group in bo.merkle_proof(nodes, idx, h):
            assert group[i & 15] == cur
            cur = bo.poseidon(K, [0] + list(group))[0]
            i >>= 4
        assert cur == nodes[-1]


# ------------------------------------------------------------------------------------------- C++ host mirror (xlp_host_bn128.hpp)
def _mini_fill(n, seed):
    M = (1 << 64) - 1
    out = np.empty(n, np.uint64)
    for i in range(n):
        z = ((seed ^ i) + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        out[i] = (z ^ (z >> 31)) % 0xFFFFFFFF00000001
    return out


@pytest.mark.gpu
@needs_constants
@pytest.mark.parametrize("h,w,seed", [(256, 12, 0x5EED0B01), (4096, 20, 0x5EED0B02), (64, 3, 0x5EED0B03)])
def test_cpp_merkle_tree_and_transcript_bn128_vs_oracle(K, h, w, seed):
    """MerkleTreeBN128 (host-pointer and GPU-resident), getGroupProof and TranscriptBN128 of host/xlp_host_bn128.hpp as
    StarkRecursiveF::genProof drives them, against the oracle"""
    import json
    import subprocess
    exe = os.path.join(ROOT, "xlayer_prover_amd", "host", "test_host_shim")
    r = subprocess.run([exe, "bn128", str(h), str(w), hex(seed)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    src = _mini_fill(h * w, seed).reshape(h, w)
    nodes = bo.merkelize(K, src)
    assert got["numNodes"] == len(nodes)
    assert int(got["root"]) == nodes[-1] and int(got["root_dev"]) == nodes[-1]
    assert got["proofs_equal"] is True
    idx = got["idx"]
    assert idx == (seed * 7 + 3) % h
    exp = [x for group in bo.merkle_proof(nodes, idx, h) for x in group]
    assert [int(x) for x in got["proof"]] == exp
    tr = bo.TranscriptBN128(K)
    tr.put([nodes[-1]])
    tr.put([int(x) for x in src.reshape(-1)[:min(w, 5)]])
    assert got["field1"] == tr.get_field()
    tr.put([int(src[0, 0])])
    assert got["field2"] == tr.get_field()
    assert got["ys"] == tr.get_permutations(8, 10)


# --------------------------------------------------------------------------------- StarkRecursiveF-shaped genProof (all stages)
@pytest.mark.gpu
@needs_constants
@pytest.mark.parametrize("nbits,steps,nq", [(5, [6, 3], 4), (9, [10, 6, 3], 8)])
def test_recursivef_shaped_genproof_gpu_equals_oracle(xlp, orc, K, nbits, steps, nq):
    """StarkRecursiveF::genProof (starkRecursiveF.cpp:150-650) is Starks::genProof with MerkleTreeBN128 trees (cm1..cm4, constants, FRI
    layers), TranscriptBN128 and 16-ary openings.  The toy-PIL flow of tests/mini_genproof_flow.py (every stage of genProof) run
    with the BN128 tree + transcript on the HIP library and on the oracle: identical roots, challenges (evaluations depend on
    them), FRI layers, query indices and openings."""
    from tests import mini_genproof_flow as flow
    from tests.mini_backends import GpuBackendBN128, OracleBackendBN128
    xlp.use_torch_stream()
    got = flow.prove(GpuBackendBN128(xlp), nbits, steps, nq, seed=nbits)
    exp = flow.prove(OracleBackendBN128(orc, bo, K), nbits, steps, nq, seed=nbits)
    assert got.keys() == exp.keys()
    for k in exp:
        assert got[k] == exp[k], k
    assert len(got["root1"]) == 1 and got["root1"][0] < bo.R_MOD and got["root1"][0] >> 64 != 0      # a BN128 root, not four GL words


# ------------------------------------------------------------------------------------------------ FFT over Fr (fflonk)
def test_oracle_fft_is_the_dft_and_inverts():
    rng = np.random.default_rng(9)
    n = 16
    a = [int.from_bytes(rng.bytes(32), "little") % bo.R_MOD for _ in range(n)]
    w = bo.fft_root(4)
    assert pow(w, n, bo.R_MOD) == 1 and pow(w, n // 2, bo.R_MOD) == bo.R_MOD - 1
    A = bo.fft(a)
    assert A == [sum(a[i] * pow(w, i * k, bo.R_MOD) for i in range(n)) % bo.R_MOD for k in range(n)]      # the definition
    assert bo.ifft(A) == a


@pytest.mark.gpu
@pytest.mark.parametrize("logn", [0, 1, 2, 5, 8, 9, 10, 12, 14, 16, 17])
def test_gpu_fft_bn254_vs_oracle(xlp, logn):
    rng = np.random.default_rng(300 + logn)
    n = 1 << logn
    a = [int.from_bytes(rng.bytes(32), "little") % bo.R_MOD for _ in range(n)]
    a[0] = bo.R_MOD - 1
    limbs = np.array([bo.to_mont_limbs(x) for x in a], dtype=np.uint64)
    got = xlp.fft_bn254(limbs)
    exp = bo.fft(a)
    assert [bo.from_mont_limbs(r) for r in got] == exp
    assert (got == np.array([bo.to_mont_limbs(x) for x in exp], dtype=np.uint64)).all()
    back = xlp.fft_bn254(got, inverse=True)
    assert (back == limbs).all()


@pytest.mark.gpu
def test_gpu_fft_bn254_2p20_properties(xlp):
    """size-independent properties at 2^20: inverse(forward(a)) == a bit for bit, linearity on a sparse probe, and A[0] = sum a[i]"""
    n = 1 << 20
    rng = np.random.default_rng(41)
    raw = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= 0x0FFFFFFFFFFFFFFF                       # < r: valid Montgomery representatives of some elements
    A = xlp.fft_bn254(raw)
    assert (xlp.fft_bn254(A, inverse=True) == raw).all()
    vals = [bo.from_mont_limbs(r) for r in raw[:: n // 64]]          # sampled check of A[0] is too weak; use a delta instead:
    d = np.zeros((n, 4), np.uint64)
    d[3] = bo.to_mont_limbs(7)
    D = xlp.fft_bn254(d)
    w = bo.fft_root(20)
    for k in (0, 1, 2, 12345, n - 1):
        assert bo.from_mont_limbs(D[k]) == 7 * pow(w, 3 * k, bo.R_MOD) % bo.R_MOD
    assert len(vals) == 64


# ------------------------------------------------------------------------------------------------ G1 MSM (fflonk commitments)
def _golden_points():
    import json
    j = json.load(open(os.path.join(ROOT, "tests", "golden", "fflonk_proof_g1_points.json")))
    return [(int(x), int(y)) for x, y in j["points"]]


def test_oracle_curve_is_pinned_by_reference_held_points():
    """the G1 commitments of the reference's own fflonk proof (testvectors/finalProof/proof.json) lie on y^2 = x^3 + 3 over the
    oracle's q; the doubling of the generator is the published 2G (EIP-196 test vectors)"""
    for p in _golden_points():
        assert bo.g1_on_curve(p) and p[0] < bo.Q_MOD
    assert bo.g1_mul(bo.G1, 2) == (1368015179489954701390400359078579693043519447331113978918064868415326638035,
                                   9918110051302171585080402603319702774565515993150576347155970296011118125764)
    assert bo.g1_mul(bo.G1, bo.R_MOD) is None                      # the group order is r
    a, b = _golden_points()[:2]
    assert bo.g1_add(bo.g1_mul(a, 5), bo.g1_mul(a, 7)) == bo.g1_mul(a, 12)
    assert bo.msm([a, b, None], [3, 4, 9]) == bo.g1_add(bo.g1_mul(a, 3), bo.g1_mul(b, 4))
    assert bo.g1_from_limbs(bo.g1_to_limbs(a)) == a and bo.g1_from_limbs([0] * 8) is None


def test_field_representation_is_pinned_by_the_reference_field_constants():
    """REFERENCE-HELD PIN of the element bytes: the reference's generated field code carries its constants as data (src/ffiasm/fr.asm / fq.asm:
    modulus q, R^2, R^3, np = -q^-1 mod 2^64; fixture tests/golden/bn254_field_constants.json).  R^2 there is (2^256)^2 mod q, so RawFr / RawFq elements
    are 4 x 64-bit Montgomery residues with R = 2^256 -- the representation the oracle converts to at its byte boundary and the kernels compute in.
    Checked: the oracle's moduli and Montgomery radix reproduce every constant; its encoding of 1 is R mod q; and the 32-bit limbs / N0INV the HIP
    kernels are compiled with (csrc/bn128.hip, csrc/bn254_msm.hip) are these numbers."""
    import json
    import re
    J = json.load(open(os.path.join(ROOT, "tests", "golden", "bn254_field_constants.json")))
    num = lambda limbs: sum(int(v, 16) << (64 * i) for i, v in enumerate(limbs))
    for name, q, mont_r in (("Fr", bo.R_MOD, bo.MONT_R), ("Fq", bo.Q_MOD, bo.Q_MONT_R)):
        c = J[name]
        assert num(c["q"]) == q
        assert num(c["half"]) == q >> 1
        assert mont_r == (1 << 256) % q
        assert num(c["R2"]) == mont_r * mont_r % q and num(c["R3"]) == pow(mont_r, 3, q)
        assert (int(c["np"][0], 16) * q + 1) % (1 << 64) == 0                      # np = -q^-1 mod 2^64
    assert bo.to_mont_limbs(1) == [(bo.MONT_R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    assert bo.from_mont_limbs([int(v, 16) for v in J["Fr"]["R2"]]) == bo.MONT_R      # R^2 read as an element is R
    assert bo.g1_to_limbs((1, 2))[:4] == [(bo.Q_MONT_R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    # the kernels' compile-time constants
    csrc = os.path.join(ROOT, "xlayer_prover_amd", "csrc")
    for fn, name in (("bn128.hip", "Fr"), ("bn254_msm.hip", "Fq")):
        src = open(os.path.join(csrc, fn)).read()
        limbs = [int(x, 16) for x in re.search(r"static constexpr u32 N\[8\] = \{([^}]*)\}", src).group(1).replace("u", "").split(",")]
        assert sum(v << (32 * i) for i, v in enumerate(limbs)) == num(J[name]["q"])
        n0inv = int(re.search(r"static constexpr u32 N0INV = (0x[0-9a-f]+)u", src).group(1), 16)
        assert n0inv == int(J[name]["np"][0], 16) & 0xFFFFFFFF


def _random_points(rng, n):
    """n curve points: small multiples of the generator and of the reference-held points, by chained additions (cheap)"""
    seeds = [bo.G1] + _golden_points()
    pts, cur = [], None
    for i in range(n):
        cur = bo.g1_add(cur, seeds[int(rng.integers(0, len(seeds)))])
        pts.append(cur)
    return pts


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 64, 300])
def test_gpu_msm_bn254_vs_oracle(xlp, n):
    rng = np.random.default_rng(500 + n)
    pts = _random_points(rng, n)
    ks = [int.from_bytes(rng.bytes(32), "little") for _ in range(n)]           # raw 256-bit scalars, not reduced
    if n >= 3:
        ks[0] = 0; ks[1] = 1; ks[2] = (1 << 256) - 1
    if n >= 17:
        pts[5] = None                                                          # a point at infinity among the bases
        pts[7] = pts[6]                                                        # equal points (doubling inside a bucket)
        ks[7] = ks[6]
        pts[9] = (pts[8][0], bo.Q_MOD - pts[8][1])                             # P and -P with the same scalar: cancels
        ks[9] = ks[8]
    bases = np.array([bo.g1_to_limbs(p) for p in pts], dtype=np.uint64).reshape(-1, 8)
    scal = np.array([[(k >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for k in ks], dtype=np.uint64).reshape(-1, 4)
    got = xlp.msm_bn254(bases, scal)
    exp = bo.msm(pts, ks)
    assert bo.g1_from_limbs(got) == exp
    assert [int(v) for v in got] == bo.g1_to_limbs(exp)


@pytest.mark.gpu
def test_gpu_msm_bn254_large_properties(xlp):
    """2^16 points (16-bit windows): linearity -- MSM(P, a) + MSM(P, b) == MSM(P, a + b) -- and MSM with all scalars equal to k ==
    k * MSM(P, 1), checked with the oracle's group law on the (small) results"""
    n = 1 << 16
    rng = np.random.default_rng(777)
    base_pts = _random_points(rng, 256)
    idx = rng.integers(0, 256, size=n)
    table = np.array([bo.g1_to_limbs(p) for p in base_pts], dtype=np.uint64)
    bases = table[idx]
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= 0x00FFFFFFFFFFFFFF
    b = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); b[:, 3] &= 0x00FFFFFFFFFFFFFF
    ab = a + b                                                      # limb-wise, no carries (62-bit limbs)
    Pa, Pb, Pab = (bo.g1_from_limbs(xlp.msm_bn254(bases, s)) for s in (a, b, ab))
    assert bo.g1_add(Pa, Pb) == Pab
    ones = np.zeros((n, 4), np.uint64); ones[:, 0] = 1
    k = 0xDEADBEEFCAFE
    ks = np.zeros((n, 4), np.uint64); ks[:, 0] = k
    S1, Sk = bo.g1_from_limbs(xlp.msm_bn254(bases, ones)), bo.g1_from_limbs(xlp.msm_bn254(bases, ks))
    assert Sk == bo.g1_mul(S1, k)
    # and against the oracle directly: sum over the 256 distinct points of (count * point)
    cnt = np.bincount(idx, minlength=256)
    assert S1 == bo.msm(base_pts, [int(c) for c in cnt])


@pytest.mark.gpu
def test_gpu_msm_bn254_wide_windows_at_the_size_of_a_commitment(xlp):
    """2^22 points: windows of 18 bits (the sizes the fflonk commitments run at use 18..22-bit windows), full-width random scalars, against
    the oracle on the 256 distinct points the bases are drawn from: sum_k (sum of the scalars that multiply point k) * P_k"""
    n = 1 << 22
    rng = np.random.default_rng(4242)
    base_pts = _random_points(rng, 256)
    idx = rng.integers(0, 256, size=n)
    table = np.array([bo.g1_to_limbs(p) for p in base_pts], dtype=np.uint64)
    bases = table[idx]
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)                           # < 2^253
    got = bo.g1_from_limbs(xlp.msm_bn254(bases, sc))
    halves = np.stack([(sc >> np.uint64(0)) & np.uint64(0xFFFFFFFF), sc >> np.uint64(32)], axis=2).reshape(n, 8)      # 32-bit digits, little-endian
    tot = [0] * 256
    for d in range(8):                                                  # exact in float64: 2^14 terms of 32 bits per point
        s = np.bincount(idx, weights=halves[:, d].astype(np.float64), minlength=256)
        for k in range(256):
            tot[k] += int(s[k]) << (32 * d)
    assert got == bo.msm(base_pts, [t % bo.R_MOD for t in tot])


@pytest.mark.gpu
@needs_constants
@pytest.mark.parametrize("nbits,steps,nq", [(6, [7, 4], 5), (9, [10, 6, 3], 9)])
def test_cpp_stark_recursivef_genproof_equals_flow(orc, K, tmp_path, nbits, steps, nq):
    """The C++ `StarkRecursiveF::genProof` (host/starkRecursiveF.hpp = StarksT<XlpHashBN128>: StarkInfo-driven, starkRecursiveF.cpp:150-650
    stage for stage) on the mini PIL: its zkin (roots as Fr strings, [query][level][16] sibling groups) equals what the Python flow
    computes on the oracle with the BN128 trees / transcript"""
    import json
    import subprocess
    from tests import mini_genproof_flow as flow
    from tests.mini_backends import OracleBackendBN128
    exe = os.path.join(ROOT, "xlayer_prover_amd", "host", "test_host_shim")
    flow.toy_stark_files(str(tmp_path), "toy", nbits, steps, nq, seed=nbits)
    r = subprocess.run([exe, "genproof_bn128", str(tmp_path), "toy"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    Z = json.loads(r.stdout)
    exp = flow.prove(OracleBackendBN128(orc, bo, K), nbits, steps, nq, seed=nbits)
    ints = lambda v: [int(x) for x in v]
    for k in ("root1", "root2", "root3", "root4", "verkey"):
        assert [int(Z[k])] == exp[k], k
    assert ints(Z["publics"]) == exp["publics"]
    assert [ints(e) for e in Z["evals"]] == exp["evals"]
    assert [x for e in Z["finalPol"] for x in ints(e)] == exp["fri"]["finalPol"]
    for si in range(1, len(steps)):
        assert [int(Z[f"s{si}_root"])] == exp["fri"]["roots"][si - 1]
        for q in range(nq):
            row, groups = exp["fri"]["si"][si - 1][q]
            assert ints(Z[f"s{si}_vals"][q]) == row and [ints(g) for g in Z[f"s{si}_siblings"][q]] == groups
    for q in range(nq):
        for k, tname in enumerate(["1", "2", "3", "4", "C"]):
            row, groups = exp["fri"]["s0"][q][k]
            assert ints(Z[f"s0_vals{tname}"][q]) == row and [ints(g) for g in Z[f"s0_siblings{tname}"][q]] == groups


@pytest.mark.gpu
@needs_constants
@pytest.mark.parametrize("h,w", [(4096, 12), (256, 21), (16, 5), (512, 48)])
def test_gpu_merkle_bn128_batched_openings_vs_oracle(xlp, K, h, w):
    """xlp_dev_merkle_bn128_proofs: every query's row + 16-node groups in one kernel == the oracle's merkle_genMerkleProof"""
    import torch
    rng = np.random.default_rng(h + w)
    src = rng.integers(0, 0xFFFFFFFF00000001, size=(h, w), dtype=np.uint64)
    d_src = xlp.from_np(src.reshape(-1))
    d_nodes = torch.zeros(xlp.merkle_bn128_num_nodes(h) * 4, dtype=torch.int64, device="cuda")
    xlp.dev_merkle_bn128(d_nodes, d_src, w, h)
    idxs = [0, h - 1, h // 2 + 1, 7 % h, 7 % h]
    got = xlp.dev_merkle_bn128_proofs(d_src, d_nodes, w, h, idxs)
    nodes = bo.merkelize(K, src)
    for q, idx in enumerate(idxs):
        assert (got[q, :w] == src[idx]).all()
        groups = bo.merkle_proof(nodes, idx, h)
        flat = got[q, w:].reshape(len(groups), 16, 4)
        assert [[bo.from_mont_limbs(r) for r in g] for g in flat] == [list(g) for g in groups]


@pytest.mark.gpu
@needs_constants
@pytest.mark.parametrize("t", [2, 3, 6, 16, 17])
def test_gpu_single_permutation_kernel_vs_oracle(xlp, K, t):
    """count <= 4 takes the one-wave-per-permutation kernel (lanes across the state: the transcript's path)"""
    rng = np.random.default_rng(900 + t)
    for count in (1, 3):
        vals = [[int.from_bytes(rng.bytes(32), "little") % bo.R_MOD for _ in range(t)] for _ in range(count)]
        st = np.array([[bo.to_mont_limbs(x) for x in row] for row in vals], dtype=np.uint64)
        got = xlp.poseidon_bn128(st, t)
        for i in range(count):
            exp = bo.poseidon(K, vals[i])
            assert (got[i] == np.array([bo.to_mont_limbs(x) for x in exp], dtype=np.uint64)).all()


@pytest.mark.gpu
@needs_constants
def test_bn128_entry_points_reject_bad_arguments(xlp):
    """error behaviour of the C ABI: non-zero status + a message, nothing computed"""
    import ctypes as C
    lib = xlp.load()
    buf = np.zeros(64, np.uint64)
    p = buf.ctypes.data_as(C.c_void_p)
    assert lib.xlp_poseidon_bn128(p, 1, 1) != 0 and b"2..17" in lib.xlp_last_error()
    assert lib.xlp_poseidon_bn128(p, 18, 1) != 0
    assert lib.xlp_merkle_bn128(p, p, 3, 3) != 0 and b"power of two" in lib.xlp_last_error()
    assert lib.xlp_fft_bn254(p, 3, 0) != 0 and b"power of two" in lib.xlp_last_error()
    assert lib.xlp_msm_bn254(None, p, p, 1) != 0
    assert lib.xlp_bn128_init(b"/nonexistent/poseidon_bn128.bin") != 0 and b"not found" in lib.xlp_last_error()
    assert lib.xlp_bn128_init(None) == 0                                     # the tables next to the library still load
    assert lib.xlp_merkle_bn128(p, p, 3, 0) == 0 and lib.xlp_fft_bn254(p, 0, 0) == 0      # empty inputs are no-ops
    assert xlp.msm_bn254(np.zeros((0, 8), np.uint64), np.zeros((0, 4), np.uint64)).tolist() == [0] * 8   # the empty sum is the point at infinity


@pytest.mark.gpu
def test_gpu_msm_bn254_skewed_scalars_heavy_buckets(xlp):
    """all scalars equal (one bucket per window holds every point) and scalars with only the top bits set (the short top window):
    buckets with more than 256 points are summed by a workgroup each; results against the oracle's group law"""
    rng = np.random.default_rng(31)
    n = 3000
    base_pts = _random_points(rng, 64)
    idx = rng.integers(0, 64, size=n)
    table = np.array([bo.g1_to_limbs(p) for p in base_pts], dtype=np.uint64)
    bases = table[idx]
    cnt = np.bincount(idx, minlength=64)
    S1 = bo.msm(base_pts, [int(c) for c in cnt])
    for k in (5, (1 << 255) + 77, (0xF << 252)):
        sc = np.array([[(k >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]] * n, dtype=np.uint64)
        got = bo.g1_from_limbs(xlp.msm_bn254(bases, sc))
        assert got == bo.g1_mul(S1, k % bo.R_MOD)


@pytest.mark.gpu
@needs_constants
def test_cpp_stark_recursivef_loads_bctree_bn128_tree(tmp_path):
    """`bctree` with verificationHashType BN128 writes <name>.consttree; StarkRecursiveF loads it (constants' extension + 16-ary tree) instead
    of computing them: the proof is the same either way"""
    import json
    import subprocess
    from tests import mini_genproof_flow as flow
    nbits, steps, nq = 7, [8, 5], 6
    flow.toy_stark_files(str(tmp_path), "toy", nbits, steps, nq, seed=nbits)
    shim = os.path.join(ROOT, "xlayer_prover_amd", "host", "test_host_shim")
    bct = os.path.join(ROOT, "xlayer_prover_amd", "host", "bctree")
    r0 = subprocess.run([shim, "genproof_bn128", str(tmp_path), "toy"], capture_output=True, text=True)
    assert r0.returncode == 0, r0.stderr[-2000:]
    (tmp_path / "ss.json").write_text(json.dumps({"nBits": nbits, "nBitsExt": steps[0], "verificationHashType": "BN128", "steps": [{"nBits": b} for b in steps]}))
    rb = subprocess.run([bct, "-c", str(tmp_path / "toy.const"), "-s", str(tmp_path / "ss.json"), "-t", str(tmp_path / "toy.consttree_bn128"),
                         "-v", str(tmp_path / "toy.verkey_bn128.json")], capture_output=True, text=True)
    assert rb.returncode == 0, rb.stderr
    r1 = subprocess.run([shim, "genproof_bn128", str(tmp_path), "toy"], capture_output=True, text=True)
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert r1.stdout == r0.stdout
    assert json.loads(r1.stdout)["verkey"] == json.load(open(tmp_path / "toy.verkey_bn128.json"))["constRoot"]


@pytest.mark.gpu
def test_gpu_msm_resident_ceiling_kernel_adds_what_it_says(xlp):
    """the measurement aid behind bench.py's msm_bn254.roofline (xlp_dev_msm_resident): every lane ends at (iters + 1) * P, Jacobian"""
    import torch
    P = _golden_points()[0]
    pt = xlp.from_np(np.array(bo.g1_to_limbs(P), dtype=np.uint64))
    lanes, iters = 130, 5
    out = torch.zeros(lanes * 12, dtype=torch.int64, device="cuda")
    xlp._ck(xlp.load().xlp_dev_msm_resident(xlp._dp(out), xlp._dp(pt), lanes, iters), "xlp_dev_msm_resident")
    jac = xlp.to_np(out).reshape(lanes, 3, 4)

    def coord(w):
        return sum(int(v) << (64 * i) for i, v in enumerate(w)) * bo.Q_MONT_RINV % bo.Q_MOD

    for i in (0, 1, 63, 64, 129):
        X, Y, Z = (coord(jac[i, k]) for k in range(3))
        zi = pow(Z, -1, bo.Q_MOD)
        assert ((X * zi * zi) % bo.Q_MOD, (Y * zi * zi * zi) % bo.Q_MOD) == bo.g1_mul(P, iters + 1)
    assert 2 <= xlp.load().xlp_msm_bn254_window_bits(1 << 22) <= 22


@pytest.mark.gpu
def test_gpu_fft_bn254_twiddle_table_is_kept_and_shared_between_sizes(xlp):
    """xlp_[dev_]fft_bn254 keeps its table of twiddles in scratch slot 8: a transform after a LARGER one reads the larger table with a
    stride, a transform after a release / a lend rebuilds it -- all equal the oracle"""
    import torch
    rng = np.random.default_rng(77)

    def check(logn):
        n = 1 << logn
        a = [int.from_bytes(rng.bytes(32), "little") % bo.R_MOD for _ in range(n)]
        limbs = np.array([bo.to_mont_limbs(x) for x in a], dtype=np.uint64)
        got = xlp.fft_bn254(limbs)
        assert [bo.from_mont_limbs(r) for r in got] == bo.fft(a)
        assert (xlp.fft_bn254(got, inverse=True) == limbs).all()

    xlp.scratch_release(1 << 8)
    check(6)                                    # builds the 2^6 table
    check(11)                                   # grows it
    check(6); check(9); check(1)                # smaller domains on the 2^11 table (9: stages through HBM)
    region = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")          # 512 KiB: holds the 2^11 table (32 KiB), not a 2^15 one
    xlp.scratch_lend(8, region)
    check(10)                                   # rebuilt inside the lent region
    assert int((region != 0).sum()) > 0
    check(15)                                   # does not fit the region: the slot's own scratch
    check(10)
    xlp.scratch_lend(8, None)
    check(7)
    xlp.scratch_release(1 << 8)
    check(8)
