"""Parity tests proper: the HIP path (through the C ABI of libxlp_gl.so) against
the CPU oracle on identical seeded inputs, plus the committed golden fixtures.
Bit-exact: Goldilocks is an integer field."""
import json
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN_PROOFS

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def xlp():
    from xlayer_prover_amd import capi
    capi.init(0)          # raises if the HIP library or device is missing: no fallback
    return capi


def ints(x):
    if isinstance(x, list):
        return [ints(y) for y in x]
    return int(x)


def edge_matrix(orc, n, ncols, seed):
    """seeded input with non-canonical and boundary words mixed in"""
    a = orc.fill(n * ncols, seed).reshape(n, ncols).copy()
    specials = np.array([0, 1, P - 1, P, P + 1, 2**64 - 1, 2**32 - 1, 2**32, 0xFFFFFFFF00000000], np.uint64)
    flat = a.reshape(-1)
    k = min(flat.size, specials.size)
    flat[:k] = specials[:k]
    return a


# ------------------------------------------------------------------ Poseidon
def test_perm_kats(xlp, orc, golden_dir):
    J = json.load(open(os.path.join(golden_dir, "poseidon_kats.json")))
    ins = np.array([ints(k["in"]) for k in J["kats"]], np.uint64)
    out = xlp.perm_batch(ins)
    for o, k in zip(out, J["kats"]):
        assert [int(x) for x in o[:4]] == ints(k["out4"])
    assert [int(x) for x in xlp.perm([0] * 12)] == ints(J["perm_zero"])
    # full 12-word result vs oracle incl. non-canonical inputs
    rng = np.random.default_rng(7)
    st = rng.integers(0, 2**64, (300, 12), dtype=np.uint64)
    st[0, :] = P - 1; st[1, :] = 2**64 - 1; st[2, :] = P
    got = xlp.perm_batch(st)
    for i in range(st.shape[0]):
        assert (got[i] == orc.perm(st[i])).all()


@pytest.mark.parametrize("w,h", [(0, 8), (1, 8), (3, 4), (4, 16), (5, 16), (8, 64), (9, 32), (16, 2), (17, 1),
                                 (18, 256), (39, 128), (751, 64), (168, 32), (6, 1024), (12, 4096)])
def test_merkle_vs_oracle(xlp, orc, w, h):
    src = edge_matrix(orc, h, w, 0x5EED0003) if w else np.zeros(0, np.uint64)
    got = xlp.merkle(src, w, h)
    exp = orc.merkletree(src, w, h)
    assert got.size == (2 * h - 1) * 4
    assert (got == exp).all()
    if w:
        assert (xlp.linear_hash(src, w, h).reshape(-1) == exp[:4 * h]).all()


def test_merkle_rejects_non_pow2(xlp):
    with pytest.raises(xlp.XlpError):
        xlp.merkle(np.zeros(3 * 5, np.uint64), 5, 3)


@pytest.mark.parametrize("name", [g[0] for g in GOLDEN_PROOFS])
def test_merkle_golden_paths(xlp, orc, golden_dir, name):
    """leaf rows + sibling paths of the reference's golden proofs hash to the proof's roots
    when the leaf and every node are computed by the HIP kernels"""
    F = json.load(open(os.path.join(golden_dir, name)))
    for Q in F["queries"]:
        for tn, root in (("1", F["root1"]), ("3", F["root3"]), ("4", F["root4"]), ("C", F["constRoot"])):
            row = np.array(ints(Q[f"s0_vals{tn}"]), np.uint64)
            cur = xlp.linear_hash(row, row.size, 1)[0]
            idx = Q["idx"]
            for sib in ints(Q[f"s0_siblings{tn}"]):
                pair = list(sib) + list(cur) if idx & 1 else list(cur) + list(sib)
                cur = xlp.perm(pair + [0] * 4)[:4]
                idx >>= 1
            assert [int(x) for x in cur] == ints(root)


# ----------------------------------------------------------------------- NTT
@pytest.mark.parametrize("lg", list(range(0, 15)))
@pytest.mark.parametrize("ncols", [1, 3, 8])
def test_ntt_vs_oracle_sizes(xlp, orc, lg, ncols):
    n = 1 << lg
    src = edge_matrix(orc, n, ncols, 0x5EED0001 + lg)
    assert (xlp.ntt(src, n, ncols) == orc.ntt(src, n, ncols)).all()
    assert (xlp.ntt(src, n, ncols, inverse=True) == orc.ntt(src, n, ncols, inverse=True)).all()


@pytest.mark.parametrize("lg,ncols", [(10, 2), (10, 5), (11, 6), (12, 7), (13, 9), (12, 16), (12, 17), (9, 751), (16, 3), (17, 8), (18, 6)])
def test_ntt_vs_oracle_widths(xlp, orc, lg, ncols):
    n = 1 << lg
    src = orc.fill(n * ncols, 0x5EED0011 + lg).reshape(n, ncols)
    assert (xlp.ntt(src, n, ncols) == orc.ntt(src, n, ncols)).all()
    assert (xlp.ntt(src, n, ncols, inverse=True) == orc.ntt(src, n, ncols, inverse=True)).all()


@pytest.mark.parametrize("lg,ncols", [(17, 1), (17, 3), (18, 5), (19, 8), (20, 6), (21, 2), (17, 70)])
def test_ntt_fast_path_vs_oracle(xlp, orc, lg, ncols):
    """two-kernel stand-alone NTT / INTT path (2^17..2^23 points), incl. in place"""
    n = 1 << lg
    src = edge_matrix(orc, n, ncols, 0x5EED0071 + lg)
    assert (xlp.ntt(src, n, ncols) == orc.ntt(src, n, ncols)).all()
    buf = src.copy()
    assert (xlp.ntt(buf, n, ncols, inverse=True, inplace=True) == orc.ntt(src, n, ncols, inverse=True)).all()


def test_ntt_inplace_and_roundtrip(xlp, orc):
    n, ncols = 1 << 14, 3
    src = orc.fill(n * ncols, 77).reshape(n, ncols)
    buf = src.copy()
    out = xlp.ntt(buf, n, ncols, inplace=True)      # INTT(LEv, LEv, ...) style, starks.cpp:326
    assert (out == orc.ntt(src, n, ncols)).all()
    back = xlp.ntt(out, n, ncols, inverse=True)
    assert (back == src).all()


def test_ntt_rejects_non_pow2(xlp):
    with pytest.raises(xlp.XlpError):
        xlp.ntt(np.zeros(12, np.uint64), 12, 1)


def test_ntt_empty(xlp):
    assert xlp.ntt(np.zeros(0, np.uint64), 0, 0).size == 0


# ----------------------------------------------------------------------- LDE
@pytest.mark.parametrize("lg", [0, 1, 2, 3, 4, 5, 7, 8, 10, 11, 12, 13])
@pytest.mark.parametrize("ext", [1, 2, 3])
def test_extend_vs_oracle(xlp, orc, lg, ext):
    n, n_ext = 1 << lg, 1 << (lg + ext)
    for ncols in (1, 3, 8, 11):
        src = edge_matrix(orc, n, ncols, 0x5EED0002 + lg)
        assert (xlp.extend(src, n_ext, n, ncols) == orc.extend_pol(src, n_ext, n, ncols)).all()


@pytest.mark.parametrize("lg,ext,ncols", [(14, 1, 18), (15, 1, 8), (16, 1, 3), (17, 3, 2), (12, 0, 4), (9, 1, 751), (10, 2, 168)])
def test_extend_vs_oracle_more(xlp, orc, lg, ext, ncols):
    n, n_ext = 1 << lg, 1 << (lg + ext)
    src = orc.fill(n * ncols, 0x5EED0022).reshape(n, ncols)
    assert (xlp.extend(src, n_ext, n, ncols) == orc.extend_pol(src, n_ext, n, ncols)).all()


@pytest.mark.parametrize("lg,ext,ncols", [(17, 1, 8), (17, 3, 70), (18, 2, 11), (19, 1, 9), (20, 1, 8), (21, 1, 4), (22, 1, 5), (17, 2, 4), (17, 0, 8)])
def test_extend_fast_path_vs_oracle(xlp, orc, lg, ext, ncols):
    """three-kernel LDE path (2^17 <= N <= 2^23, ncols >= 4): all column-chunk / tail-mask shapes"""
    n, n_ext = 1 << lg, 1 << (lg + ext)
    src = edge_matrix(orc, n, ncols, 0x5EED0042 + lg)
    assert (xlp.extend(src, n_ext, n, ncols) == orc.extend_pol(src, n_ext, n, ncols)).all()


@pytest.mark.parametrize("lg,ext,ncols", [(22, 1, 13), (22, 2, 8), (23, 1, 9)])
def test_extend_2p22_2p23_vs_oracle(xlp, orc, lg, ext, ncols):
    """N >= 2^22 against the oracle: full 8-column tiles, ragged tiles, non-canonical input words"""
    n, n_ext = 1 << lg, 1 << (lg + ext)
    src = edge_matrix(orc, n, ncols, 0x5EED0062 + lg)
    assert (xlp.extend(src, n_ext, n, ncols) == orc.extend_pol(src, n_ext, n, ncols)).all()


def test_lde_config2_device_resident_properties(xlp, orc):
    """BASELINE config 2 at full size (2^23 x 8 -> 2^24 x 8) through the device API:
    (a) one whole column against the oracle, (b) size-independent properties:
    even output rows of a coset LDE restricted to the original domain... here:
    INTT_ext(out) must have zero upper half and lower half = shift^i * INTT_n(in)."""
    import torch
    lg, ncols = 23, 8
    n, n_ext = 1 << lg, 1 << (lg + 1)
    src = torch.empty(n * ncols, dtype=torch.int64, device="cuda")
    out = torch.empty(n_ext * ncols, dtype=torch.int64, device="cuda")
    xlp.dev_fill(src, 0x5EED0002, 0)
    xlp.dev_extend(out, src, n_ext, n, ncols)
    xlp.sync()
    h_src = xlp.to_np(src).reshape(n, ncols)
    assert (h_src[:4].reshape(-1) == orc.fill(4 * ncols, 0x5EED0002)).all()   # same synthetic trace as the oracle's
    h_out = xlp.to_np(out).reshape(n_ext, ncols)
    col = 5
    exp = orc.extend_pol(h_src[:, col].copy(), n_ext, n, 1).reshape(-1)
    assert (h_out[:, col] == exp).all()
    # property on all columns: coefficients of the extension vanish above n
    co = torch.empty_like(out)
    xlp.dev_ntt(co, out, n_ext, ncols, inverse=True)
    ci = torch.empty_like(src)
    xlp.dev_ntt(ci, src, n, ncols, inverse=True)
    xlp.sync()
    h_co = xlp.to_np(co).reshape(n_ext, ncols)
    assert not h_co[n:].any()
    h_ci = xlp.to_np(ci).reshape(n, ncols)
    for i in (0, 1, 2, 12345, n - 1):
        s = orc.pow_(7, i)
        for c in range(ncols):
            assert int(h_co[i, c]) == orc.mul(int(h_ci[i, c]), s)
    # the VALUES of every column, by an algorithm without a transform: input and output are evaluations of one polynomial of degree < n, so their barycentric
    # interpolants (oracle, orc_barycentric_eval) agree at random points of the field -- at the full size, all eight columns
    import random
    rng = random.Random(23)
    for c in range(ncols):
        z = rng.randrange(0xFFFFFFFF00000001)
        assert orc.barycentric_eval(h_src.ravel()[c:], n, ncols, 1, z) == orc.barycentric_eval(h_out.ravel()[c:], n_ext, ncols, orc.shift(), z), c
    # and the first rows of a second, different-seed run differ (no stale buffers)
    assert h_out.any()


def test_ntt_2p20_single_column_config1(xlp, orc):
    """BASELINE config 1: NTT 2^20 x 1 (the reference's CPU-runnable plumbing case)."""
    n = 1 << 20
    src = orc.fill(n, 0x5EED0001).reshape(n, 1)
    assert (xlp.ntt(src, n, 1) == orc.ntt(src, n, 1)).all()


# ----------------------------------------------------------------------- FRI
@pytest.mark.parametrize("bits,red", [(4, 0), (4, 1), (6, 2), (8, 3), (10, 4), (12, 5), (16, 4), (20, 4)])
def test_fri_fold_vs_oracle(xlp, orc, bits, red):
    n = 1 << bits
    pol = orc.fill(3 * n, 0x5EED0006 + bits).reshape(n, 3)
    x = orc.fill(3, 99)
    sinv = orc.inv(7)
    for _ in range(bits % 3):
        sinv = orc.mul(sinv, sinv)
    got = xlp.fri_fold(pol, bits, red, x, sinv)
    assert (got == orc.fri_fold(pol, bits, red, x, sinv)).all()


def test_fri_fold_golden(xlp, orc, golden_dir):
    """fold relations of the reference's golden proofs, computed by the HIP fold kernel:
    embed the query's group values into an otherwise random polynomial."""
    for name, _ in GOLDEN_PROOFS:
        F = json.load(open(os.path.join(golden_dir, name)))
        steps = F["steps_nBits"]
        for Q in F["queries"][:2]:
            for si in range(1, 5):
                bits, red = steps[si - 1], steps[si - 1] - steps[si]
                M = 1 << steps[si]
                g = Q["idx"] % M
                pol = orc.fill(3 << bits, 1234 + si).reshape(-1, 3)
                vals = np.array(ints(Q[f"s{si}_vals"]), np.uint64).reshape(-1, 3)
                pol[g::M] = vals
                x = np.array(ints(F["special_x"][si]), np.uint64)
                got = xlp.fri_fold(pol, bits, red, x, int(F["shift_inv_per_step"][si]))[g]
                if si < 4:
                    nb = steps[si + 1]
                    exp = ints(Q[f"s{si+1}_vals"])[3 * (g >> nb): 3 * (g >> nb) + 3]
                else:
                    exp = ints(F["finalPol"][g])
                assert [int(v) for v in got] == exp


@pytest.mark.parametrize("n,tb", [(64, 2), (1 << 12, 8), (1 << 16, 12), (96 * 32, 5), (1 << 10, 0), (1 << 10, 10)])
def test_transpose3_vs_oracle(xlp, orc, n, tb):
    if n % (1 << tb):
        pytest.skip("2^tb must divide n")
    src = orc.fill(3 * n, 5).reshape(n, 3)
    assert (xlp.transpose3(src, n, tb) == orc.transpose3(src, n, tb)).all()


# --------------------------------------------------------------- field layer
def _field_inputs():
    rng = np.random.default_rng(11)
    sp = np.array([0, 1, 2, P - 2, P - 1, P, P + 1, 2**64 - 1, 2**64 - 2, 2**32 - 1, 2**32, 2**32 + 1, 2**63, 2**63 - 1,
                   0xFFFFFFFF00000000, 0xFFFFFFFEFFFFFFFF, 0x00000000FFFFFFFF, 0xFFFFFFFF, 1 << 48, (1 << 48) - 1], np.uint64)
    a, b = np.meshgrid(sp, sp)
    a = np.concatenate([a.reshape(-1), rng.integers(0, 2**64, 4000, dtype=np.uint64)])
    b = np.concatenate([b.reshape(-1), rng.integers(0, 2**64, 4000, dtype=np.uint64)])
    return a, b


def test_field_ops_exhaustive_edges(xlp):
    a, b = _field_inputs()
    ai = [int(x) for x in a]
    bi = [int(x) for x in b]
    exp = {
        0: [(x % P + y % P) % P for x, y in zip(ai, bi)],
        1: [(x % P - y % P) % P for x, y in zip(ai, bi)],
        2: [(x * y) % P for x, y in zip(ai, bi)],
        3: [x % P for x in ai],
        4: [(x + y) % P for x, y in zip(ai, bi)],
        5: [(x - y) % P for x, y in zip(ai, bi)],
    }
    for op, e in exp.items():
        got = xlp.field_op(a, b, op)
        assert [int(v) for v in got] == e, f"field op {op}"
    for k in (1, 12, 24, 32, 36, 48, 60, 64, 72, 84, 95):
        got = xlp.field_op(a, b, 100 + k)
        assert [int(v) for v in got] == [(x << k) % P for x in ai], f"mul_pow2c<{k}>"
    for k in (0, 1, 31, 32, 33, 63, 64, 65, 95):
        got = xlp.field_op(a, b, 200 + k)
        assert [int(v) for v in got] == [(x << k) % P for x in ai], f"mul_pow2({k})"


# ---------------------------------------------------------------- multi-GPU plumbing on one GPU
def test_gpu_hash_pairs_and_world1_commit(xlp, orc):
    import torch
    import torch.distributed as dist
    from xlayer_prover_amd import shard
    nodes = orc.fill(16 * 4, 77).reshape(16, 4)
    got = xlp.to_np(shard.gpu_hash_pairs(xlp.from_np(nodes)))
    exp = np.stack([orc.perm(list(nodes[2 * i]) + list(nodes[2 * i + 1]) + [0] * 4)[:4] for i in range(8)])
    assert (got == exp).all()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, n_ext, ncols = 1 << 10, 1 << 11, 9
        full = orc.fill(n * ncols, 0x5EED0004).reshape(n, ncols)
        root, nodes_t, rows = shard.sharded_commit(xlp.from_np(full), n, n_ext, ncols)
        xlp.sync()
        ext = orc.extend_pol(full, n_ext, n, ncols)
        assert (xlp.to_np(rows) == ext).all()
        assert (xlp.to_np(nodes_t) == orc.merkletree(ext, ncols, n_ext)).all()
        assert (xlp.to_np(root) == orc.merkletree(ext, ncols, n_ext)[-4:]).all()
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------- row-scale helpers
@pytest.mark.parametrize("n", [1, 3, 4, 5, 1000, (1 << 14) + 7])
def test_batch_inverse_vs_oracle(xlp, orc, n):
    src = orc.fill(3 * n, 0x5EED0061).reshape(n, 3)
    if n > 4:
        src[2] = 0                       # a zero element maps to zero and does not poison its batch
        src[3] = [0, 5, 0]
    t = xlp.from_np(src.reshape(-1))
    out = xlp.from_np(np.zeros(3 * n, np.uint64))
    xlp.dev_batch_inverse(out, t, n, 3); xlp.sync()
    got = xlp.to_np(out).reshape(n, 3)
    assert (got == orc.batch_inverse(src, 3)).all()
    for i in (0, n - 1):
        assert list(orc.mul3(got[i], src[i])) == [1, 0, 0]
    # in place + dim 1
    xlp.dev_batch_inverse(t, t, n, 3); xlp.sync()
    assert (xlp.to_np(t).reshape(n, 3) == got).all()
    b = orc.fill(n, 0x5EED0062)
    tb = xlp.from_np(b); xlp.dev_batch_inverse(tb, tb, n, 1); xlp.sync()
    assert (xlp.to_np(tb) == orc.batch_inverse(b, 1)).all()


@pytest.mark.parametrize("n", [1, 2, 9, 2048, 2049, 5000, (1 << 16) + 3])
def test_calculate_z_vs_oracle(xlp, orc, n):
    num = orc.fill(3 * n, 0x5EED0063).reshape(n, 3)
    den = orc.fill(3 * n, 0x5EED0064).reshape(n, 3)
    z = xlp.from_np(np.zeros(3 * n, np.uint64))
    xlp.dev_calculate_z(z, xlp.from_np(num.reshape(-1)), xlp.from_np(den.reshape(-1)), n); xlp.sync()
    assert (xlp.to_np(z).reshape(n, 3) == orc.calculate_z(num, den)).all()


def test_calculate_z_closes_for_a_permutation(xlp, orc):
    """grand-product property the reference asserts (polinomial.hpp:603-606): if den is a permutation of num,
    z[n-1] * num[n-1] / den[n-1] == 1 -- at 2^20 rows, beyond what the sequential oracle does quickly"""
    n = 1 << 20
    num = orc.fill(3 * n, 0x5EED0065).reshape(n, 3)
    den = np.roll(num, 12345, axis=0).copy()
    z = xlp.from_np(np.zeros(3 * n, np.uint64))
    xlp.dev_calculate_z(z, xlp.from_np(num.reshape(-1)), xlp.from_np(den.reshape(-1)), n); xlp.sync()
    zl = xlp.to_np(z).reshape(n, 3)[-1]
    chk = orc.mul3(zl, orc.mul3(num[-1], orc.inv3(den[-1])))
    assert list(chk) == [1, 0, 0]
    assert list(xlp.to_np(z)[:3]) == [1, 0, 0]


def test_evmap_vs_oracle(xlp, orc):
    nbits, eb = 10, 1
    n, ne = 1 << nbits, 1 << (nbits + eb)
    LEv = orc.fill(3 * n, 0x5EED0066).reshape(n, 3)
    LpEv = orc.fill(3 * n, 0x5EED0067).reshape(n, 3)
    secA = orc.fill(ne * 18, 0x5EED0068).reshape(ne, 18)     # a committed section of 18 base columns
    secB = orc.fill(ne * 9, 0x5EED0069).reshape(ne, 9)       # 3 ext-3 columns
    ents = [(secA, 0, 1, False), (secA, 17, 1, True), (secA, 5, 1, False), (secB, 0, 3, False), (secB, 6, 3, True), (secA, 5, 1, True)]
    exp = orc.evmap(LEv, LpEv, n, eb, ents)
    tA, tB = xlp.from_np(secA.reshape(-1)), xlp.from_np(secB.reshape(-1))
    dev_ents = [((tA if e[0] is secA else tB), e[0].shape[1], e[1], e[2], e[3]) for e in ents]
    ev = xlp.from_np(np.zeros(3 * len(ents), np.uint64))
    xlp.dev_evmap(ev, xlp.from_np(LEv.reshape(-1)), xlp.from_np(LpEv.reshape(-1)), n, eb, dev_ents); xlp.sync()
    assert (xlp.to_np(ev).reshape(-1, 3) == exp).all()


# ---------------------------------------------------------------- full-size properties
def test_ntt_2p24_generic_path_roundtrip_and_column(xlp, orc):
    """largest transform of a zkEVM proof (q polynomial, starks.cpp:262,285): 2^24 x 3, device resident:
    INTT(NTT(x)) == x on everything, one whole column against the oracle"""
    import torch
    n, ncols = 1 << 24, 3
    src = torch.empty(n * ncols, dtype=torch.int64, device="cuda")
    xlp.dev_fill(src, 0x5EED0081, 0)
    f = torch.empty_like(src); b = torch.empty_like(src)
    xlp.dev_ntt(f, src, n, ncols, inverse=False)
    xlp.dev_ntt(b, f, n, ncols, inverse=True)
    xlp.sync()
    assert torch.equal(b, src)
    h_src = xlp.to_np(src).reshape(n, ncols)
    col = 1
    exp = orc.ntt(h_src[:, col].copy(), n, 1).reshape(-1)
    assert (xlp.to_np(f).reshape(n, ncols)[:, col] == exp).all()


def test_merkle_large_subtree_property(xlp, orc):
    """2^22 x 8 tree (device resident): (a) sampled leaves and their paths verify against the root with the oracle,
    (b) the root equals the tree over the four quarter-trees' roots (perfect binary tree over natural row order --
    the property the multi-GPU cap assembly relies on)"""
    import torch
    h, w = 1 << 22, 8
    src = torch.empty(h * w, dtype=torch.int64, device="cuda")
    xlp.dev_fill(src, 0x5EED0082, 0)
    nodes = torch.empty((2 * h - 1) * 4, dtype=torch.int64, device="cuda")
    xlp.dev_merkle(nodes, src, w, h)
    q = h // 4
    qroots = []
    for i in range(4):
        qn = torch.empty((2 * q - 1) * 4, dtype=torch.int64, device="cuda")
        xlp.dev_merkle(qn, src[i * q * w:(i + 1) * q * w], w, q)
        xlp.sync()
        qroots.append(xlp.to_np(qn[-4:]))
    xlp.sync()
    root = xlp.to_np(nodes[-4:])
    l = [orc.perm(list(qroots[0]) + list(qroots[1]) + [0] * 4)[:4], orc.perm(list(qroots[2]) + list(qroots[3]) + [0] * 4)[:4]]
    assert list(orc.perm(list(l[0]) + list(l[1]) + [0] * 4)[:4]) == list(root)
    h_nodes = xlp.to_np(nodes)
    for idx in (0, 1, 123456, h - 1):
        row = orc.fill(w, 0x5EED0082, idx * w)
        assert list(h_nodes[4 * idx:4 * idx + 4]) == list(orc.linear_hash(row))
        sib, off, cnt, i = [], 0, h, idx
        while cnt > 1:
            sib.append(h_nodes[off + (i ^ 1) * 4: off + (i ^ 1) * 4 + 4]); off += 4 * cnt; cnt //= 2; i >>= 1
        assert list(orc.merkle_root_from_path(row, np.concatenate(sib), idx)) == list(root)


def _plookup_inputs(orc, n, dim, seed, distinct):
    """t: table with duplicates, f: every value drawn from t (heavily skewed)"""
    rng = np.random.default_rng(seed)
    base = orc.fill(distinct * dim, seed).reshape(distinct, dim)
    t = base[np.sort(rng.integers(0, distinct, n))]
    pick = np.minimum((rng.random(n) ** 4 * n).astype(np.int64), n - 1)     # skew: most f hit few rows
    f = t[pick]
    return np.ascontiguousarray(f), np.ascontiguousarray(t)


@pytest.mark.parametrize("n,dim,distinct", [(1, 1, 1), (8, 1, 3), (8, 3, 8), (1000, 1, 37), (4096, 3, 500), (1 << 16, 1, 1 << 12), (1 << 16, 3, 1 << 16), (1 << 20, 1, 1 << 19), (1 << 20, 3, 50)])
def test_calculate_h1h2_vs_oracle(xlp, orc, n, dim, distinct):
    f, t = _plookup_inputs(orc, n, dim, 0x5EED0090 + n, distinct)
    exp1, exp2 = orc.calculate_h1h2(f, t, dim)
    h1 = xlp.from_np(np.zeros(n * dim, np.uint64)); h2 = xlp.from_np(np.zeros(n * dim, np.uint64))
    xlp.dev_calculate_h1h2(h1, h2, xlp.from_np(f.reshape(-1)), xlp.from_np(t.reshape(-1)), n, dim); xlp.sync()
    assert (xlp.to_np(h1) == exp1.reshape(-1)).all() and (xlp.to_np(h2) == exp2.reshape(-1)).all()


@pytest.mark.parametrize("n", [1, 9, 2049, 1 << 16])
def test_calculate_z_on_section_views_vs_oracle(xlp, orc, n):
    """Polinomial views (polinomial.hpp:24-40): num, den and z are column groups of row-major sections of different widths; the
    words around z must stay untouched"""
    wn, wd, wz = 7, 12, 5
    secn = orc.fill(n * wn, 0x5EED00A1).reshape(n, wn); secd = orc.fill(n * wd, 0x5EED00A2).reshape(n, wd)
    secz = orc.fill(n * wz, 0x5EED00A3).reshape(n, wz)
    tn, td, tz = xlp.from_np(secn.reshape(-1)), xlp.from_np(secd.reshape(-1)), xlp.from_np(secz.reshape(-1))
    xlp.dev_calculate_z_strided(tz[2:], wz, tn[4:], wn, td[1:], wd, n); xlp.sync()
    exp = secz.copy(); exp[:, 2:5] = orc.calculate_z(np.ascontiguousarray(secn[:, 4:7]), np.ascontiguousarray(secd[:, 1:4]))
    assert (xlp.to_np(tz).reshape(n, wz) == exp).all()
    with pytest.raises(xlp.XlpError, match="stride"):
        xlp.dev_calculate_z_strided(tz, 2, tn, wn, td, wd, n)


@pytest.mark.parametrize("n,dim,distinct", [(8, 1, 3), (4096, 3, 500), (1 << 16, 1, 1 << 12), (1 << 16, 3, 1 << 16)])
def test_calculate_h1h2_on_section_views_vs_oracle(xlp, orc, n, dim, distinct):
    f, t = _plookup_inputs(orc, n, dim, 0x5EED00B0 + n, distinct)
    exp1, exp2 = orc.calculate_h1h2(f, t, dim)
    wf, wt, wh = dim + 4, dim + 9, 2 * dim + 3
    secf = orc.fill(n * wf, 0x5EED00B1).reshape(n, wf); secf[:, 2:2 + dim] = f.reshape(n, dim)
    sect = orc.fill(n * wt, 0x5EED00B2).reshape(n, wt); sect[:, 9:9 + dim] = t.reshape(n, dim)
    sech = orc.fill(n * wh, 0x5EED00B3).reshape(n, wh)
    tf, tt, th = xlp.from_np(secf.reshape(-1)), xlp.from_np(sect.reshape(-1)), xlp.from_np(sech.reshape(-1))
    xlp.dev_calculate_h1h2_strided(th[1:], wh, th[1 + dim:], wh, tf[2:], wf, tt[9:], wt, n, dim); xlp.sync()
    exp = sech.copy(); exp[:, 1:1 + dim] = exp1.reshape(n, dim); exp[:, 1 + dim:1 + 2 * dim] = exp2.reshape(n, dim)
    assert (xlp.to_np(th).reshape(n, wh) == exp).all()


def test_calculate_h1h2_number_not_included(xlp, orc):
    t = np.arange(1, 65, dtype=np.uint64); f = t.copy(); f[10] = 1000
    h = xlp.from_np(np.zeros(64, np.uint64)); h2 = xlp.from_np(np.zeros(64, np.uint64))
    with pytest.raises(xlp.XlpError, match="not included"):
        xlp.dev_calculate_h1h2(h, h2, xlp.from_np(f), xlp.from_np(t), 64, 1)
    with pytest.raises(ValueError):
        orc.calculate_h1h2(f, t, 1)


@pytest.mark.parametrize("n,dim", [(1, 1), (7, 1), (1 << 12, 1), ((1 << 20) + 17, 1), (5, 3), (1 << 12, 3), (700001, 3)])
def test_geometric_tables_vs_oracle(xlp, orc, n, dim):
    """x_n / x_2ns / x (starks.hpp:149-183) and LEv / LpEv (starks.cpp:318-322): out[k] = first * ratio^k"""
    first = orc.fill(dim, 900 + n); ratio = orc.fill(dim, 901 + n)
    if dim == 1 and n == 1 << 12:
        first, ratio = np.array([orc.shift()], np.uint64), np.array([orc.w(12)], np.uint64)     # x_2ns of a 2^12 domain
    out = xlp.from_np(np.zeros(n * dim, np.uint64))
    xlp.dev_geometric(out, n, dim, first, ratio); xlp.sync()
    assert (xlp.to_np(out) == orc.geometric(n, dim, first, ratio).reshape(-1)).all()


@pytest.mark.parametrize("n", [1, 3, 4, 1000, (1 << 18) + 5])
def test_x_div_x_sub_xi_vs_oracle(xlp, orc, n):
    """xDivXSubXi (starks.cpp:340-362), including a point equal to xi (zero difference -> zero)"""
    x = orc.fill(n, 0x5EED0700 + n)
    xi = orc.fill(3, 77)
    if n >= 4:
        xi[1] = 0; xi[2] = 0; x[2] = xi[0]
    out = xlp.from_np(np.zeros(3 * n, np.uint64))
    xlp.dev_x_div_x_sub_xi(out, xlp.from_np(x), xi, n); xlp.sync()
    assert (xlp.to_np(out).reshape(-1, 3) == orc.x_div_x_sub_xi(x, xi)).all()


@pytest.mark.parametrize("nbits,ebits,qdeg,qdim", [(4, 1, 2, 3), (10, 1, 2, 3), (8, 2, 3, 3), (12, 1, 1, 1)])
def test_q_split_vs_oracle(xlp, orc, nbits, ebits, qdeg, qdim):
    n, ne = 1 << nbits, 1 << (nbits + ebits)
    qq1 = orc.fill(ne * qdim, 0x5EED0800 + nbits)
    shift_in = orc.pow_(orc.inv(orc.shift()), n)
    out = xlp.from_np(np.full(ne * qdeg * qdim, 7, np.uint64))
    xlp.dev_q_split(out, xlp.from_np(qq1), n, ne, qdeg, qdim, shift_in); xlp.sync()
    assert (xlp.to_np(out) == orc.q_split(qq1, n, ne, qdeg, qdim, shift_in)).all()


@pytest.mark.parametrize("w,h", [(0, 8), (3, 1), (5, 2), (18, 64), (751, 1 << 10)])
def test_merkle_proofs_batched_vs_oracle(xlp, orc, w, h):
    """xlp_dev_merkle_proofs (getGroupProof for many indices in one call) against the oracle's proof walk"""
    src = orc.fill(max(w * h, 1), 0x5EED0B00 + w)[: w * h]
    nodes = orc.merkletree(src, w, h)
    d_src = xlp.from_np(src if w * h else np.zeros(1, np.uint64)); d_nodes = xlp.from_np(nodes)
    idxs = sorted({0, h - 1, h // 2, (h * 3) // 7})
    got = xlp.dev_merkle_proofs(d_src, d_nodes, w, h, idxs)
    for row, idx in zip(got, idxs):
        assert (row == orc.merkle_proof(nodes, src, w, h, idx)).all()
    with pytest.raises(xlp.XlpError, match="out of range"):
        xlp.dev_merkle_proofs(d_src, d_nodes, w, h, [h])


@pytest.mark.parametrize("nbits,eb,width,nev,seed", [(3, 1, 5, 7, 1), (12, 1, 37, 300, 2), (13, 3, 21, 64, 3), (14, 1, 751, 900, 4)])
def test_evmap_many_entries_vs_oracle(xlp, orc, nbits, eb, width, nev, seed):
    """the entry-per-lane evmap kernel: more entries than one block, rows not a multiple of the staged sub-chunk, mixed
    dimensions and prime flags, duplicate columns, non-canonical inputs, a second (narrow) section"""
    rng = np.random.default_rng(seed)
    n, ne = 1 << nbits, 1 << (nbits + eb)
    LEv = orc.fill(3 * n, 0x5EED0E00 + seed).reshape(n, 3)
    LpEv = orc.fill(3 * n, 0x5EED0E10 + seed).reshape(n, 3)
    secA = rng.integers(0, 2**64, size=(ne, width), dtype=np.uint64)          # any 64-bit words
    secB = orc.fill(ne * 6, 0x5EED0E20 + seed).reshape(ne, 6)
    ents = []
    for _ in range(nev):
        if rng.integers(0, 5) == 0:
            ents.append((secB, int(rng.integers(0, 2)) * 3, 3, bool(rng.integers(0, 2))))
        elif width >= 3 and rng.integers(0, 6) == 0:
            ents.append((secA, int(rng.integers(0, width - 2)), 3, bool(rng.integers(0, 2))))
        else:
            ents.append((secA, int(rng.integers(0, width)), 1, bool(rng.integers(0, 2))))
    exp = orc.evmap(LEv, LpEv, n, eb, ents)
    tA, tB = xlp.from_np(secA.reshape(-1)), xlp.from_np(secB.reshape(-1))
    dev_ents = xlp.evmap_entries([((tA if e[0] is secA else tB), e[0].shape[1], e[1], e[2], e[3]) for e in ents])
    ev = xlp.from_np(np.zeros(3 * len(ents), np.uint64))
    xlp.dev_evmap(ev, xlp.from_np(LEv.reshape(-1)), xlp.from_np(LpEv.reshape(-1)), n, eb, dev_ents); xlp.sync()
    assert (xlp.to_np(ev).reshape(-1, 3) == exp).all()


# ---------------------------------------------------------------- dot_columns (linear part of the FRI polynomial)
@pytest.mark.parametrize("nrows,ngroups,seed,reps", [(1, 1, 1, 1), (255, 2, 2, 1), (256, 3, 3, 1), (1000, 4, 4, 1), (5000, 3, 5, 1), (300, 2, 6, 40), (70, 1, 7, 60)])
def test_dot_columns_vs_bigint(xlp, orc, nrows, ngroups, seed, reps):
    """xlp_dev_dot_columns against Python integers (the extension product restated as in oracle mul3): several sections of
    different widths, base and extension columns, repeated columns, arbitrary 64-bit words as inputs.  reps > 1: thousands of entries
    per group -- more than one carry-free accumulator takes (1 024 products), and more per 64-word window than the plan allows a
    group (512), so the accumulators are folded mid-way and windows are split; the all-ones rows make every partial product maximal"""
    import torch
    rng = np.random.default_rng(seed)
    Pm = 0xFFFFFFFF00000001
    widths = [37, 3, 130, 18]
    secs_h = [rng.integers(0, 2**64, size=nrows * w, dtype=np.uint64) for w in widths]      # non-canonical words included
    if reps > 1:
        for a, w in zip(secs_h, widths):
            a[:w] = 2**64 - 1                                                                   # row 0: every word all ones
    secs = [xlp.from_np(a) for a in secs_h]
    ncoef = 50
    coefs_h = orc.fill(3 * ncoef, 0x5EED0D07 + seed)
    coefs_h[0:3] = [2**64 - 1, Pm, Pm + 5]                                                  # non-canonical coefficient words
    ents = []
    for si, w in enumerate(widths):
        for _ in range(max(2, w // 2) * reps):
            dim = 3 if (w >= 3 and rng.integers(0, 4) == 0) else 1
            ents.append((si, w, int(rng.integers(0, w - dim + 1)), dim, int(rng.integers(0, ngroups)), int(rng.integers(0, ncoef)) if reps == 1 else int(rng.integers(0, 3))))
    out = torch.zeros(nrows * 3 * ngroups, dtype=torch.int64, device="cuda")
    xlp.dev_dot_columns(out, [(secs[si], w, c, d, g, k) for si, w, c, d, g, k in ents], ngroups, xlp.from_np(coefs_h), nrows)
    xlp.sync()
    got = xlp.to_np(out).reshape(nrows, ngroups, 3)

    def mul3(a, b):
        a0, a1, a2 = a; b0, b1, b2 = b
        return [(a0 * b0 + a1 * b2 + a2 * b1) % Pm, (a0 * b1 + a1 * b0 + a1 * b2 + a2 * b1 + a2 * b2) % Pm, (a0 * b2 + a2 * b0 + a1 * b1 + a2 * b2) % Pm]
    for row in list(range(min(nrows, 3))) + [nrows - 1] + [int(r) for r in rng.integers(0, nrows, 6)]:
        exp = [[0, 0, 0] for _ in range(ngroups)]
        for si, w, c, d, g, k in ents:
            x = [int(v) for v in secs_h[si][row * w + c: row * w + c + d]] + [0] * (3 - d)
            m = mul3([int(v) for v in coefs_h[3 * k:3 * k + 3]], x)
            exp[g] = [(u + v) % Pm for u, v in zip(exp[g], m)]
        assert [[int(v) for v in r] for r in got[row]] == exp, row


@pytest.mark.parametrize("nblocks,seed", [(1, 1), (2, 2), (7, 3), (739, 4)])
def test_perm_chain_equals_sequential_permutations(xlp, orc, nblocks, seed):
    """xlp_perm_chain (Transcript::put of a long message as one call, transcript.cpp:12-29) against block-by-block hash_full_result on the
    oracle: elements and the starting state are arbitrary 64-bit words; 739 blocks = the evaluations of a zkEVM proof"""
    rng = np.random.default_rng(seed)
    elems = rng.integers(0, 2**64, size=8 * nblocks, dtype=np.uint64)
    state = rng.integers(0, 2**64, size=4, dtype=np.uint64)
    got = xlp.perm_chain(state, elems)
    cap = state.copy()
    out = None
    for b in range(nblocks):
        out = orc.perm(np.concatenate([elems[8 * b:8 * b + 8], cap]))
        cap = out[:4]
    assert [int(v) for v in got] == [int(v) for v in out]


def test_dot_columns_without_entries_gives_zeros(xlp):
    import torch
    co = torch.zeros(6, dtype=torch.int64, device="cuda")
    out = torch.full((100 * 3 * 2,), 7, dtype=torch.int64, device="cuda")
    xlp.dev_dot_columns(out, [], 2, co, 100); xlp.sync()
    assert int(out.abs().sum().item()) == 0
    xlp.dev_dot_columns(out, [], 2, co, 0); xlp.sync()                 # no rows: nothing to do


def test_dot_columns_rejections(xlp):
    import torch
    sec = torch.zeros(64, dtype=torch.int64, device="cuda"); co = torch.zeros(6, dtype=torch.int64, device="cuda")
    out = torch.zeros(8 * 3, dtype=torch.int64, device="cuda")
    for ent, ng, msg in [((sec, 8, 0, 2, 0, 0), 1, "dim must be 1 or 3"), ((sec, 8, 0, 1, 1, 0), 1, "group out of range"),
                         ((sec, 8, 0, 1, 0, 2), 1, "coefficient index"), ((sec, 8, 0, 1, 0, 0), 5, "at most 4 groups")]:
        with pytest.raises(xlp.XlpError, match=msg):
            xlp.dev_dot_columns(out, [ent], ng, co, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("n,ws,wd,dim", [(1, 3, 3, 3), (1000, 7, 3, 3), (4097, 389, 3, 3), (1 << 12, 3, 168, 3), (513, 5, 9, 1)])
def test_copy_cols_strided_views(xlp, n, ws, wd, dim):
    """xlp_dev_copy_cols = Polinomial::copy between strided views (the transposeH1H2 / transposeZ helpers, starks.cpp:406-555):
    only the addressed columns of the destination change"""
    import torch
    rng = np.random.default_rng(n + ws)
    src = rng.integers(0, 1 << 63, size=n * ws, dtype=np.int64)
    dst = rng.integers(0, 1 << 63, size=n * wd, dtype=np.int64)
    cs, cd = int(rng.integers(0, ws - dim + 1)), int(rng.integers(0, wd - dim + 1))
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    xlp.dev_copy_cols(d_dst[cd:], wd, d_src[cs:], ws, n, dim)
    xlp.sync()
    exp = dst.reshape(n, wd).copy()
    exp[:, cd:cd + dim] = src.reshape(n, ws)[:, cs:cs + dim]
    assert (d_dst.cpu().numpy().reshape(n, wd) == exp).all()


# ---------------------------------------------------------------- band-pipelined commit (xlp_dev_commit / xlp_dev_commit_from_host)
@pytest.mark.parametrize("w,bands", [(5, [5]), (8, [8]), (9, [8, 1]), (70, [64, 6]), (751, [64] * 11 + [47]), (168, [8, 16, 24, 120]), (23, [16, 7])])
def test_linear_hash_absorbed_in_bands_equals_linear_hash(xlp, orc, w, bands):
    """the leaf sponge continued band by band (xlp_dev_linear_hash_absorb) == PoseidonGoldilocks::linear_hash of the whole row"""
    import torch
    h = 300
    src = edge_matrix(orc, h, w, 0x5EED00B1)
    d = xlp.from_np(src.reshape(-1))
    st = torch.empty(4 * h, dtype=torch.int64, device="cuda"); st.fill_(-1)
    c0 = 0
    for i, wc in enumerate(bands):
        xlp.dev_linear_hash_absorb(st, d, w, c0, wc, h, i == 0, i == len(bands) - 1); c0 += wc
    xlp.sync()
    exp = np.array([orc.linear_hash(r) for r in src])
    assert (xlp.to_np(st).reshape(h, 4) == exp).all()
    with pytest.raises(xlp.XlpError):
        xlp.dev_linear_hash_absorb(st, d, w, 4, 4, h, True, False)            # a band may not start inside a sponge block


@pytest.mark.parametrize("lg,ext,w,first,wc", [(10, 1, 70, 8, 40), (12, 2, 17, 5, 12), (17, 1, 75, 64, 11), (17, 1, 130, 0, 64), (18, 1, 9, 3, 5)])
def test_extend_column_window_vs_oracle(xlp, orc, lg, ext, w, first, wc):
    """extendPol of a window of columns inside row-major sections (xlp_dev_extend_cols): the window equals the oracle's LDE of
    those columns, everything outside it is untouched"""
    import torch
    n, n_ext = 1 << lg, 1 << (lg + ext)
    src = orc.fill(n * w, 0x5EED00B2 + lg).reshape(n, w)
    d_in = xlp.from_np(src.reshape(-1))
    d_out = torch.full((n_ext * w,), 7, dtype=torch.int64, device="cuda")
    xlp.dev_extend_cols(d_out, d_in, n_ext, n, w, first, wc)
    xlp.sync()
    got = xlp.to_np(d_out).reshape(n_ext, w)
    exp = orc.extend_pol(np.ascontiguousarray(src[:, first:first + wc]), n_ext, n, wc)
    assert (got[:, first:first + wc] == exp).all()
    assert (got[:, :first] == 7).all() and (got[:, first + wc:] == 7).all()


@pytest.mark.parametrize("lg,ext,w", [(9, 1, 3), (10, 1, 70), (12, 1, 751), (17, 1, 75), (14, 2, 40), (11, 3, 64), (13, 1, 200), (10, 1, 320), (10, 1, 327), (9, 2, 408), (8, 1, 1003)])
def test_band_pipelined_commit_from_host_vs_oracle(xlp, orc, lg, ext, w):
    """starks.cpp:53-60 as ONE pipelined call on a witness that lives in host memory (prover.cpp:435): column bands are uploaded
    on a second stream while the previous band is extended and absorbed into the leaf hashes.  The uploaded section, its
    extension and the whole node buffer equal the oracle's extendPol + merkletree; the device-resident form (xlp_dev_commit)
    and a pageable (not page-locked) source give the same."""
    import torch
    n, n_ext = 1 << lg, 1 << (lg + ext)
    host = xlp.HostBuffer(n * w)
    host.array[:] = edge_matrix(orc, n, w, 0x5EED00B3 + lg).reshape(-1)
    src = host.array.reshape(n, w).copy()
    sec_n = torch.zeros(n * w, dtype=torch.int64, device="cuda"); sec_2ns = torch.zeros(n_ext * w, dtype=torch.int64, device="cuda")
    nodes = torch.zeros((2 * n_ext - 1) * 4, dtype=torch.int64, device="cuda")
    ms = xlp.dev_commit_from_host(sec_n, sec_2ns, nodes, host, n_ext, n, w, want_upload_ms=True)
    xlp.sync()
    assert ms >= 0.0
    exp_ext = orc.extend_pol(src, n_ext, n, w)
    exp_nodes = orc.merkletree(exp_ext, w, n_ext)
    assert (xlp.to_np(sec_n).reshape(n, w) == src).all()
    assert (xlp.to_np(sec_2ns).reshape(n_ext, w) == exp_ext).all()
    assert (xlp.to_np(nodes) == exp_nodes).all()
    # device-resident form
    sec_2ns.zero_(); nodes.zero_()
    xlp.dev_commit(sec_2ns, nodes, sec_n, n_ext, n, w); xlp.sync()
    assert (xlp.to_np(sec_2ns).reshape(n_ext, w) == exp_ext).all() and (xlp.to_np(nodes) == exp_nodes).all()
    # pageable source
    sec_n.zero_(); nodes.zero_()
    xlp.dev_commit_from_host(sec_n, sec_2ns, nodes, np.ascontiguousarray(src.reshape(-1)), n_ext, n, w); xlp.sync()
    assert (xlp.to_np(nodes) == exp_nodes).all() and (xlp.to_np(sec_n).reshape(n, w) == src).all()
    host.close()


def test_band_pipelined_commit_2p22_x_751_equals_plain_commit(xlp, orc):
    """a zkEVM-width section, 2^22 rows (25 GB of witness in page-locked memory, 12 bands in flight): the pipelined commit equals
    upload + xlp_dev_extend + xlp_dev_merkle on the same data bit for bit (both checked against the oracle at small sizes above),
    the uploaded section equals the host rows, and the upload ran at a PCIe-like rate"""
    import torch
    lg, w = 22, 751
    n, n_ext = 1 << lg, 1 << (lg + 1)
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 120e9:
        pytest.skip("needs ~110 GB of free HBM")
    seed = 0x5EED00B4
    tmp = torch.empty(n * w, dtype=torch.int64, device="cuda")
    xlp.dev_fill(tmp, seed, 0)
    host = xlp.HostBuffer(n * w)
    xlp.copy_d2h(host, tmp)                                     # the "executor" leaves the witness in host memory
    assert (host.array[:8] == orc.fill(8, seed)).all() and (host.array[-8:] == orc.fill(8, seed, n * w - 8)).all()
    ref_2ns = torch.empty(n_ext * w, dtype=torch.int64, device="cuda"); ref_nodes = torch.empty((2 * n_ext - 1) * 4, dtype=torch.int64, device="cuda")
    xlp.dev_extend(ref_2ns, tmp, n_ext, n, w); xlp.dev_merkle(ref_nodes, ref_2ns, w, n_ext); xlp.sync()
    ref_root = xlp.to_np(ref_nodes[-4:]).copy()
    del ref_2ns
    torch.cuda.empty_cache()
    sec_n = torch.zeros(n * w, dtype=torch.int64, device="cuda"); sec_2ns = torch.empty(n_ext * w, dtype=torch.int64, device="cuda")
    nodes = torch.zeros((2 * n_ext - 1) * 4, dtype=torch.int64, device="cuda")
    ms = xlp.dev_commit_from_host(sec_n, sec_2ns, nodes, host, n_ext, n, w, want_upload_ms=True)
    xlp.sync()
    assert torch.equal(sec_n, tmp)
    assert torch.equal(nodes, ref_nodes)
    assert [int(v) for v in xlp.to_np(nodes[-4:])] == [int(v) for v in ref_root]
    gbs = n * w * 8 / ms / 1e6
    assert gbs > 20.0, f"upload ran at {gbs:.1f} GB/s"
    host.close()


# ---------------------------------------------------------------- host-pointer drop-in: pipelined + device-copy cache
@pytest.mark.parametrize("lg,ext,w", [(12, 1, 70), (17, 1, 8), (13, 2, 200)])      # (17: the three-kernel fast path behind the host-pointer entry; narrow: the oracle is the cost)
def test_host_pointer_extend_then_merkle_with_cache(xlp, orc, lg, ext, w):
    """the reference's call pair on HOST pointers -- ntt.extendPol(cm_2ns, cm_n, ...) then merkelize(cm_2ns) (starks.cpp:53-56) --
    through xlp_extend / xlp_merkle: banded three-stream pipeline, and with the cache on the second call finds the extended
    section on the device.  Results equal the oracle's with the cache off, on, after an invalidation, and after the host
    changed the section and said so."""
    n, n_ext = 1 << lg, 1 << (lg + ext)
    src = edge_matrix(orc, n, w, 0x5EED00C1 + lg)
    exp_ext = orc.extend_pol(src, n_ext, n, w)
    exp_nodes = orc.merkletree(exp_ext, w, n_ext)
    for cache in (0, 1 << 34):
        xlp.cache_enable(cache)
        got = xlp.extend(src, n_ext, n, w)
        assert (got == exp_ext).all()
        held = xlp.cache_bytes()
        assert (held > 0) == bool(cache)
        nodes = xlp.merkle(got, w, n_ext)                       # same host array object -> same address range
        assert (nodes == exp_nodes).all()
        if cache:
            assert xlp.cache_bytes() >= got.nbytes
            # the host modifies the section and tells the library: the stale device copy must not be used
            got[5, 3] ^= np.uint64(1)
            xlp.cache_invalidate(got)
            exp2 = orc.merkletree(got, w, n_ext)
            assert (xlp.merkle(got, w, n_ext) == exp2).all()
            # an extend INTO a cached range drops the old copy of it
            got2 = got                                          # reuse the buffer as the output of another extension
            src2 = orc.fill(n * w, 0x5EED00C2).reshape(n, w)
            out = xlp.extend_into(got2, src2, n_ext, n, w)
            assert (out == orc.extend_pol(src2, n_ext, n, w)).all()
            assert (xlp.merkle(out, w, n_ext) == orc.merkletree(out, w, n_ext)).all()
    xlp.cache_enable(0)
    assert xlp.cache_bytes() == 0


def test_host_pointer_outputs_drop_stale_device_copies(xlp, orc):
    """every host-pointer entry point that WRITES a host range drops the device copies of it (include/xlp_gl.h, xlp_cache_enable): proof 1
    commits a section (xlp_merkle caches its device copy), proof 2 rewrites the same host buffer through another call -- xlp_ntt(dst = the
    section), xlp_batch_inverse, xlp_transpose3 -- and commits it again: the second tree must be the tree of the NEW content."""
    n, w = 1 << 10, 12
    xlp.cache_enable(1 << 30)
    keep = []          # the sections stay allocated: a buffer freed and handed out again by the allocator with other content is a host range
    try:               # changed behind the library's back -- the caller's xlp_cache_invalidate, not what this test is about (it failed on exactly that once)
        for rewrite in ("ntt", "batch_inverse", "transpose3"):
            sec = np.ascontiguousarray(orc.fill(n * w, 0x5EED00D1).reshape(n, w))
            keep.append(sec)
            assert (xlp.merkle(sec, w, n) == orc.merkletree(sec, w, n)).all()
            assert xlp.cache_bytes() >= sec.nbytes                          # the section's device copy is now cached under its host range
            src = orc.fill(n * w, 0x5EED00D2).reshape(n, w)
            lib = xlp.load()
            if rewrite == "ntt":
                xlp._ck(lib.xlp_ntt(xlp._p(sec), xlp._p(src), n, w, 1), "xlp_ntt")
                assert (sec == orc.ntt(src, n, w, inverse=True)).all()
            elif rewrite == "batch_inverse":
                xlp._ck(lib.xlp_batch_inverse(xlp._p(sec), xlp._p(src), n * w // 3, 3), "xlp_batch_inverse")
            else:
                xlp._ck(lib.xlp_transpose3(xlp._p(sec), xlp._p(src), n * w // 3, 4), "xlp_transpose3")
            assert (xlp.merkle(sec, w, n) == orc.merkletree(sec, w, n)).all(), f"stale device copy hashed after {rewrite}"
    finally:
        xlp.cache_enable(0)
