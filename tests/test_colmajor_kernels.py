"""Column-major device sections (include/xlp_gl.h, csrc/colmajor.hip, ntt_lde_cm.hip, poseidon.hip): every stage of the commit path on the
column-major / coset-major layout against the oracle's row-major restatement on the same data -- the transposes, extendPol (the three
column-major kernels and the staged fallback), the leaf sponge in bands, the tree, the openings, and the two band-pipelined commits."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def xlp():
    import torch
    from xlayer_prover_amd import capi
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.init(0)
    capi.use_torch_stream()
    return capi


def _cm(a, ext):
    from tools import zkevm_geometry as G
    return G.matrix_to_colmajor(a, ext)


def _rows(flat, rows, w, ext):
    from tools import zkevm_geometry as G
    return G.matrix_from_colmajor(flat, rows, w, ext)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,w,ext,col0,ncols", [(64, 8, 0, 0, 8), (1 << 10, 70, 1, 3, 65), (1 << 9, 5, 3, 1, 4), (96, 130, 0, 0, 130), (1 << 12, 13, 2, 0, 13)])
def test_transposes_between_the_two_layouts(xlp, orc, rows, w, ext, col0, ncols):
    import torch
    a = orc.fill(rows * w, 0x5EED0D00 + rows + w).reshape(rows, w)
    src = xlp.from_np(a.reshape(-1))
    dst = torch.zeros(ncols * rows, dtype=torch.int64, device="cuda")
    xlp.dev_rows_to_cm(dst, rows, src, w, col0, ncols, rows, ext)
    xlp.sync()
    assert (xlp.to_np(dst) == _cm(a[:, col0:col0 + ncols], ext)).all()
    back = torch.zeros(rows * w, dtype=torch.int64, device="cuda")
    xlp.dev_cm_to_rows(back, w, col0, dst, rows, ncols, rows, ext)
    xlp.sync()
    exp = np.zeros_like(a); exp[:, col0:col0 + ncols] = a[:, col0:col0 + ncols]
    assert (xlp.to_np(back).reshape(rows, w) == exp).all()


@pytest.mark.gpu
def test_rows_to_cm_canonicalises(xlp, orc):
    """what enters a column-major section is canonical (XLP_EXPR_CANONICAL, include/xlp_gl.h): words >= p of a row-major image leave the transposition
    reduced, everything else as it is"""
    import torch
    P = 0xFFFFFFFF00000001
    rows, w = 256, 7
    a = orc.fill(rows * w, 0x5EED0D77).reshape(rows, w).copy()
    a[3, 2] = P; a[100, 6] = 2**64 - 1; a[255, 0] = P + 5
    dst = torch.zeros(w * rows, dtype=torch.int64, device="cuda")
    xlp.dev_rows_to_cm(dst, rows, xlp.from_np(a.reshape(-1)), w, 0, w, rows, 1)
    xlp.sync()
    assert (xlp.to_np(dst) == _cm(a % np.uint64(P), 1)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n0,n1,n", [(5, 2, 3000), (7, 4, 1), (3, 1, 255), (2, 8, 1024), (1, 1, 1025)])
def test_copy_3d_regroups_runs(xlp, orc, n0, n1, n):
    """the packing of a sharded commit on column-major windows: [columns][cosets][a rank's rows] out of whole columns and back, with the
    window's halo row behind every coset block (stride n + 1); short runs (the halo itself: n = 1) take the per-word kernel"""
    import numpy as np
    import torch
    full = 3 * n + 5                                                     # a coset of the whole column; the rank's rows start at n + 2
    a = orc.fill(n0 * n1 * full, 0x5EED0D30).reshape(n0, n1, full)
    src = xlp.from_np(a.reshape(-1))
    packed = torch.zeros(n0 * n1 * n, dtype=torch.int64, device="cuda")
    xlp.dev_copy_3d(packed, n1 * n, n, src, n1 * full, full, n0, n1, n, src_off=n + 2)
    xlp.sync()
    assert (xlp.to_np(packed).reshape(n0, n1, n) == a[:, :, n + 2:2 * n + 2]).all()
    win = torch.full((n0 * n1 * (n + 1),), 7, dtype=torch.int64, device="cuda")
    xlp.dev_copy_3d(win, n1 * (n + 1), n + 1, packed, n1 * n, n, n0, n1, n)
    xlp.sync()
    w = xlp.to_np(win).reshape(n0, n1, n + 1)
    assert (w[:, :, :n] == a[:, :, n + 2:2 * n + 2]).all() and (w[:, :, n] == np.uint64(7)).all()


@pytest.mark.gpu
def test_copy_view_between_layouts(xlp, orc):
    import torch
    n = 1000
    a = orc.fill(n * 7, 0x5EED0D10).reshape(n, 7)                       # row-major section of width 7: the ext-3 polynomial at columns 2..4
    cm = xlp.from_np(_cm(a, 0))
    dense = torch.zeros(n * 3, dtype=torch.int64, device="cuda")
    xlp.dev_copy_view(dense, 3, 1, cm, 1, n, n, 3, src_off=2 * n)       # column-major view (row stride 1, component stride n) -> dense [n][3]
    xlp.sync()
    assert (xlp.to_np(dense).reshape(n, 3) == a[:, 2:5]).all()
    back = torch.zeros(n * 7, dtype=torch.int64, device="cuda")
    xlp.dev_copy_view(back, 1, n, dense, 3, 1, n, 3, dst_off=4 * n)
    xlp.sync()
    assert (_rows(xlp.to_np(back), n, 7, 0)[:, 4:7] == a[:, 2:5]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("lg,ext,w", [(4, 1, 3), (10, 1, 9), (12, 2, 5), (17, 1, 5), (17, 3, 2), (18, 2, 9), (19, 1, 33), (20, 1, 3), (13, 4, 4)])
def test_extend_cm_equals_oracle(xlp, orc, lg, ext, w):
    """2^17.. rows with blow-up <= 8: the three column-major kernels; the others: the staged fallback"""
    import torch
    n, ne = 1 << lg, 1 << (lg + ext)
    a = orc.fill(n * w, 0x5EED0D20 + lg).reshape(n, w)
    a[0, 0] = 0xFFFFFFFFFFFFFFFF; a[1 % n, 0] = 0xFFFFFFFF00000001                       # non-canonical words
    exp = orc.extend_pol(a, ne, n, w)
    src = xlp.from_np(_cm(a, 0))
    out = torch.zeros(ne * w, dtype=torch.int64, device="cuda")
    xlp.dev_extend_cm(out, ne, src, n, ne, n, w)
    xlp.sync()
    got = _rows(xlp.to_np(out), ne, w, ext)
    assert (got == exp).all(), np.argwhere(got != exp)[:5]


@pytest.mark.gpu
def test_extend_cm_at_the_size_of_config2_by_the_barycentric_formula(xlp, orc):
    """the column-major extendPol (what genBatchProof runs) at BASELINE configs[1]'s size, 2^23 x 8 -> 2^24 x 8: the VALUES of all eight columns against an
    algorithm without a transform -- the input column on <w_N> and the output column (coset-major storage back to natural order) on shift <w_Next> have the same
    barycentric interpolant at random points (oracle, orc_barycentric_eval)"""
    import random
    import torch
    lg, w = 23, 8
    n, ne = 1 << lg, 2 << lg
    src = torch.empty(n * w, dtype=torch.int64, device="cuda")
    xlp.dev_fill(src, 0x5EED0D99, 0)                                   # [column][row]: column c is the words c * n .. (c + 1) * n of the fill
    out = torch.zeros(ne * w, dtype=torch.int64, device="cuda")
    xlp.dev_extend_cm(out, ne, src, n, ne, n, w)
    xlp.sync()
    h_in = xlp.to_np(src).reshape(w, n)
    h_out = xlp.to_np(out).reshape(w, 2, n)                           # coset-major: [column][coset r][q], natural row (q << 1) | r
    rng = random.Random(99)
    for c in range(w):
        nat = np.ascontiguousarray(h_out[c].T).reshape(-1)
        z = rng.randrange(0xFFFFFFFF00000001)
        assert orc.barycentric_eval(h_in[c], n, 1, 1, z) == orc.barycentric_eval(nat, ne, 1, orc.shift(), z), c


@pytest.mark.gpu
@pytest.mark.parametrize("lg,ext,w", [(8, 1, 751), (10, 2, 20), (9, 1, 4), (9, 1, 3), (7, 3, 9), (6, 0, 16), (9, 1, 0)])
def test_merkle_cm_equals_oracle_and_openings(xlp, orc, lg, ext, w):
    import torch
    h = 1 << (lg + ext)
    a = orc.fill(h * w, 0x5EED0D30 + w).reshape(h, w)
    exp = orc.merkletree(a, w, h)
    src = xlp.from_np(_cm(a, ext)) if w else torch.zeros(1, dtype=torch.int64, device="cuda")
    nodes = torch.zeros((2 * h - 1) * 4, dtype=torch.int64, device="cuda")
    xlp.dev_merkle_cm(nodes, src, h, w, h, ext)
    xlp.sync()
    assert (xlp.to_np(nodes) == exp).all()
    if w > 8:
        # the same leaves absorbed in three bands (the pipelined commit's shape)
        st = torch.zeros((2 * h - 1) * 4, dtype=torch.int64, device="cuda")
        b1, b2 = 8, (w // 16) * 8 + 8
        xlp.dev_linear_hash_absorb_cm(st, src, h, 0, b1, h, ext, True, False)
        xlp.dev_linear_hash_absorb_cm(st, src, h, b1, b2 - b1, h, ext, False, b2 == w)
        if b2 < w:
            xlp.dev_linear_hash_absorb_cm(st, src, h, b2, w - b2, h, ext, False, True)
        xlp.sync()
        assert (xlp.to_np(st)[:4 * h] == exp[:4 * h]).all()
    if w:
        idxs = [0, 1, h - 1, h // 2 + 1, 5 % h]
        got = xlp.dev_merkle_proofs_cm(src, h, nodes, w, h, ext, idxs)
        ref = xlp.dev_merkle_proofs(xlp.from_np(a.reshape(-1)), nodes, w, h, idxs)
        assert (got == ref).all()


@pytest.mark.gpu
@pytest.mark.parametrize("lg,ext,w,from_host", [(10, 1, 751, True), (10, 1, 168, False), (11, 2, 78, True), (9, 1, 4, True), (9, 1, 70, True), (17, 1, 70, True), (17, 1, 9, False)])
def test_commit_cm_equals_oracle(xlp, orc, lg, ext, w, from_host):
    import torch
    n, ne = 1 << lg, 1 << (lg + ext)
    a = orc.fill(n * w, 0x5EED0D40 + w + lg).reshape(n, w)
    e = orc.extend_pol(a, ne, n, w)
    exp_nodes = orc.merkletree(e, w, ne)
    sec_n = torch.zeros(n * w, dtype=torch.int64, device="cuda")
    sec_2 = torch.zeros(ne * w, dtype=torch.int64, device="cuda")
    nodes = torch.zeros((2 * ne - 1) * 4, dtype=torch.int64, device="cuda")
    if from_host:
        stage = torch.full((n * w + 3,), 5, dtype=torch.int64, device="cuda")
        xlp.dev_commit_from_host_cm(sec_n, n, sec_2, ne, nodes, np.ascontiguousarray(a.reshape(-1)), stage, ne, n, w, clear_stage=(w % 2 == 0))
        xlp.sync()
        assert (_rows(xlp.to_np(sec_n), n, w, 0) == a).all()
        st = xlp.to_np(stage)
        # (the bands land dense, one after the other; cleared once transposed when asked for)
        assert (st[n * w:] == 5).all() and ((st[:n * w] == 0).all() if w % 2 == 0 else (st[:n * w] != 5).all())
    else:
        sec_n = xlp.from_np(_cm(a, 0))
        xlp.dev_commit_cm(sec_2, ne, nodes, sec_n, n, ne, n, w)
        xlp.sync()
    assert (_rows(xlp.to_np(sec_2), ne, w, ext) == e).all()
    assert (xlp.to_np(nodes) == exp_nodes).all()
    import ctypes as C
    l, m = C.c_double(0), C.c_double(0)
    assert xlp.load().xlp_dev_commit_last_times(C.byref(l), C.byref(m)) == 0 and l.value > 0 and m.value > 0


@pytest.mark.gpu
@pytest.mark.parametrize("nbits,eb,width,nev,seed", [(3, 1, 5, 7, 1), (12, 1, 37, 300, 2), (13, 3, 21, 64, 3), (14, 1, 751, 900, 4)])
def test_evmap_cm_equals_oracle(xlp, orc, nbits, eb, width, nev, seed):
    """Starks::evmap on column-major / coset-major sections against the oracle on the row-major ones: more entries than a block holds, rows not a
    multiple of the staged sub-chunk, mixed dimensions and prime flags, duplicate columns, arbitrary 64-bit words, two sections"""
    rng = np.random.default_rng(seed)
    n, ne = 1 << nbits, 1 << (nbits + eb)
    LEv = orc.fill(3 * n, 0x5EED0E00 + seed).reshape(n, 3)
    LpEv = orc.fill(3 * n, 0x5EED0E10 + seed).reshape(n, 3)
    secA = rng.integers(0, 2**64, size=(ne, width), dtype=np.uint64)
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
    tA, tB = xlp.from_np(_cm(secA, eb)), xlp.from_np(_cm(secB, eb))
    dev_ents = xlp.evmap_entries_cm([((tA if e[0] is secA else tB), ne, e[1], e[2], e[3]) for e in ents])
    ev = xlp.from_np(np.zeros(3 * len(ents), np.uint64))
    xlp.dev_evmap_cm(ev, xlp.from_np(LEv.reshape(-1)), xlp.from_np(LpEv.reshape(-1)), n, dev_ents); xlp.sync()
    assert (xlp.to_np(ev).reshape(-1, 3) == exp).all()


@pytest.mark.gpu
@pytest.mark.parametrize("nrows,ngroups,seed,reps", [(1, 1, 1, 1), (255, 2, 2, 1), (256, 3, 3, 1), (1000, 4, 4, 1), (5000, 3, 5, 1), (300, 2, 6, 40)])
def test_dot_columns_cm_equals_the_row_major_kernel(xlp, orc, nrows, ngroups, seed, reps):
    """xlp_dev_dot_columns_cm on the transposed sections == xlp_dev_dot_columns (itself checked against Python integers in test_gpu_parity.py):
    base and extension columns, repeated columns and repeated (column, group) pairs, arbitrary 64-bit words, > 1 024 columns per run"""
    import torch
    rng = np.random.default_rng(seed)
    Pm = 0xFFFFFFFF00000001
    widths = [37, 3, 130, 18] if reps == 1 else [37, 3, 1300, 18]
    secs_h = [rng.integers(0, 2**64, size=nrows * w, dtype=np.uint64) for w in widths]
    if reps > 1:
        for a, w in zip(secs_h, widths):
            a[:w] = 2**64 - 1
    secs = [xlp.from_np(a) for a in secs_h]
    secs_cm = [xlp.from_np(_cm(a.reshape(nrows, w), 0)) for a, w in zip(secs_h, widths)]
    ncoef = 50
    coefs_h = orc.fill(3 * ncoef, 0x5EED0D07 + seed)
    coefs_h[0:3] = [2**64 - 1, Pm, Pm + 5]
    ents = []
    for si, w in enumerate(widths):
        for _ in range(max(2, w // 2) * (reps if w < 1000 else 1)):
            dim = 3 if (w >= 3 and rng.integers(0, 4) == 0) else 1
            ents.append((si, w, int(rng.integers(0, w - dim + 1)), dim, int(rng.integers(0, ngroups)), int(rng.integers(0, ncoef)) if reps == 1 else int(rng.integers(0, 3))))
    if reps > 1:
        ents += [(2, 1300, c, 1, int(rng.integers(0, ngroups)), int(rng.integers(0, 3))) for c in range(1300)]       # every column once more: > 1 024 distinct columns
    co = xlp.from_np(coefs_h)
    ref = torch.zeros(nrows * 3 * ngroups, dtype=torch.int64, device="cuda")
    xlp.dev_dot_columns(ref, [(secs[si], w, c, d, g, k) for si, w, c, d, g, k in ents], ngroups, co, nrows)
    out = torch.full((nrows * 3 * ngroups + 5,), 7, dtype=torch.int64, device="cuda")
    xlp.dev_dot_columns_cm(out, nrows, [(secs_cm[si], nrows, c, d, g, k) for si, w, c, d, g, k in ents], ngroups, co, nrows)
    xlp.sync()
    got = xlp.to_np(out)
    assert (got[nrows * 3 * ngroups:] == 7).all()
    assert (got[:nrows * 3 * ngroups].reshape(3 * ngroups, nrows).T == xlp.to_np(ref).reshape(nrows, 3 * ngroups)).all()
