"""N > 1 path on CPU: world_size-2 (and 4) gloo processes run the partition / exchange /
cap-assembly logic of xlayer_prover_amd/shard.py with the oracle standing in for the kernels
(injected -- the product module never imports it), and the result must equal the
single-process oracle Merkle tree of the full extended trace."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, n_ext, ncols, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import gl_oracle as orc
    from xlayer_prover_amd import shard
    full = orc.fill(n * ncols, 0x5EED0004).reshape(n, ncols)
    c0, c1 = shard.col_range(ncols, world, rank)
    local = torch.from_numpy(full[:, c0:c1].copy().view(np.int64))
    as_u = lambda t: t.numpy().view(np.uint64)
    to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64))
    lde = lambda x: to_t(orc.extend_pol(as_u(x), n_ext, n, x.shape[1]))
    merkle = lambda rows: to_t(orc.merkletree(as_u(rows), rows.shape[1], rows.shape[0]))

    def hash_pairs(nodes):
        a = as_u(nodes).reshape(-1, 8)
        return to_t(np.stack([orc.perm(list(r) + [0] * 4)[:4] for r in a]))
    root, nodes, rows = shard.sharded_commit(local, n, n_ext, ncols, lde=lde, merkle=merkle, hash_pairs=hash_pairs)
    r0, r1 = shard.row_range(n_ext, world, rank)
    q.put((rank, as_u(root).tolist(), as_u(rows).tolist() if rank == 0 else None, (r0, r1), as_u(nodes[-4:]).tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ncols", [(2, 8), (2, 5), (4, 11)])
def test_sharded_commit_equals_single_process_tree(world, ncols, orc):
    n, n_ext = 1 << 6, 1 << 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world * 7 + ncols
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, n_ext, ncols, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = orc.fill(n * ncols, 0x5EED0004).reshape(n, ncols)
    ext = orc.extend_pol(full, n_ext, n, ncols)
    tree = orc.merkletree(ext, ncols, n_ext)
    exp_root = [int(v) for v in tree[-4:]]
    per = n_ext // world
    lvl = int(np.log2(per))                      # sub-roots live at this level of the reference tree
    off = sum(4 * (n_ext >> k) for k in range(lvl))
    for rank, root, rows, (r0, r1), sub in res:
        assert root == exp_root                                        # every rank derives the same root
        assert sub == [int(v) for v in tree[off + 4 * rank: off + 4 * rank + 4]]   # sub-root = node (lvl, rank)
        assert (r0, r1) == (rank * per, (rank + 1) * per)
        if rows is not None:
            assert rows == ext[r0:r1].tolist()                         # exchange delivered all columns of my rows


def _exchange_worker(rank, world, port, h, ncols, max_words, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xlayer_prover_amd import shard
    shard.MAX_MESSAGE_WORDS = max_words                        # many rounds, ragged tails (the product value is 2^25 words)
    full = torch.arange(h * ncols, dtype=torch.int64).view(h, ncols) * 7 + 3
    c0, c1 = shard.col_range(ncols, world, rank)
    r0, r1 = shard.row_range(h, world, rank)
    rows = shard.columns_to_rows(full[:, c0:c1].contiguous(), ncols)
    cols = shard.rows_to_columns(full[r0:r1].contiguous(), h)
    q.put((rank, bool((rows == full[r0:r1]).all()), bool((cols == full[:, c0:c1]).all())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ncols,max_words", [(2, 5, 7), (4, 11, 5), (4, 6, 1 << 25), (2, 8, 16)])
def test_banded_exchange_rounds(world, ncols, max_words):
    """columns <-> rows in rounds of at most MAX_MESSAGE_WORDS per message: spans of unequal width end in different rounds"""
    h = 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world * 7 + ncols
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, h, ncols, max_words, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True, True) for r in range(world)]


def test_partitions():
    from xlayer_prover_amd import shard
    for W, G in [(751, 8), (168, 8), (408, 8), (6, 4), (8, 2), (5, 2)]:
        spans = [shard.col_range(W, G, g) for g in range(G)]
        assert spans[0][0] == 0 and spans[-1][1] == W
        assert all(spans[i][1] == spans[i + 1][0] for i in range(G - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert [shard.col_range(751, 8, g)[1] - shard.col_range(751, 8, g)[0] for g in range(8)] == [94] * 7 + [93]


ROW_BUFFERS = {"pols": None, "const_pols": 5, "const_pols_2ns": 4, "x_n": 1, "x_2ns": 1, "x_div_x_sub_xi": 3, "x_div_x_sub_wxi": 3, "q_2ns": 3, "f_2ns": 3}


def _expr_worker(rank, world, port, seed, n, ninstr, q):
    """one rank of a row-sharded constraint step: windows = own rows + halo rows of the next rank; the oracle interpreter
    stands in for the HIP VM (injected here, the product module never imports it)"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import gl_oracle as orc
    from tests.expr_programs import make_buffers, random_program
    from xlayer_prover_amd import shard
    prog, width = random_program(seed, n, ninstr, prime_dst=True)
    full = make_buffers(orc, 0x5EED0C00 + seed, n, width, zi_size=4)
    r0, r1 = shard.row_range(n, world, rank)
    rows, halo = r1 - r0, 3
    widths = dict(ROW_BUFFERS); widths["pols"] = width
    win = {}
    for name, arr in full.items():
        if name in widths:
            w = widths[name]
            local = torch.from_numpy(arr.reshape(n, w)[r0:r1].copy().view(np.int64).reshape(-1))
            win[name] = np.ascontiguousarray(shard.halo_exchange(local, w, halo).numpy().view(np.uint64))
        else:
            win[name] = arr.copy()
    orc.expr_run(prog.code, prog.ntmp1, prog.ntmp3, win, 0, rows, 5, 4)       # local row numbers, global modulus in the program
    # next-row stores landed in the halo rows: hand them to the owner (columns written by pols_prime destinations)
    from xlayer_prover_amd import expr as E
    from tests import expr_real as R
    cols = sorted({(int(prog.code[qq]), dim, int(prog.code[qq + 1])) for k, s, kind, dim, qq in R.operands(prog.code) if s == 0 and kind == E.K_POLS_PRIME})
    if cols:
        t = torch.from_numpy(win["pols"].view(np.int64))
        shard.halo_writeback(t, width, rows, halo, cols)
        win["pols"] = t.numpy().view(np.uint64)
    q.put((rank, r0, r1, {k: win[k].reshape(rows + halo, widths[k])[:rows].copy() for k in ("pols", "q_2ns", "f_2ns")}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,seed", [(2, 1), (4, 2)])
def test_row_sharded_constraint_step_equals_single_process(world, seed, orc):
    """constraint steps shard by rows with a halo for the next-row operands (and the reverse exchange for
    next-row stores); the assembled result equals the single-process run over the whole domain"""
    from tests.expr_programs import make_buffers, random_program
    n, ninstr = 256, 150
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world * 13 + seed
    procs = [ctx.Process(target=_expr_worker, args=(r, world, port, seed, n, ninstr, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    prog, width = random_program(seed, n, ninstr, prime_dst=True)
    exp = make_buffers(orc, 0x5EED0C00 + seed, n, width, zi_size=4)
    before = exp["pols"].copy()
    orc.expr_run(prog.code, prog.ntmp1, prog.ntmp3, exp, 0, n, 5, 4)
    assert (exp["pols"] != before).any()
    widths = {"pols": width, "q_2ns": 3, "f_2ns": 3}
    for rank, r0, r1, out in res:
        for k, w in widths.items():
            assert (out[k] == exp[k].reshape(n, w)[r0:r1]).all(), (rank, k)


def _z_ev_worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import gl_oracle as orc
    from xlayer_prover_amd import shard
    as_u = lambda t: t.numpy().view(np.uint64)
    to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1))
    num = orc.fill(3 * n, 0x5EED0F01).reshape(n, 3); den = orc.fill(3 * n, 0x5EED0F02).reshape(n, 3)
    r0, r1 = shard.row_range(n, world, rank)
    calc_z = lambda a, b: to_t(orc.calculate_z(as_u(a), as_u(b)))

    def scale3(z, s):
        v = as_u(z).reshape(-1, 3)
        for i in range(v.shape[0]):
            v[i] = orc.mul3(v[i], s)
    z = shard.sharded_calculate_z(to_t(num[r0:r1]), to_t(den[r0:r1]), calc_z, scale3)
    # evmap: rows of LEv / LpEv and of a 2x extended section, local windows
    eb, w = 1, 7
    LEv = orc.fill(3 * n, 0x5EED0F03).reshape(n, 3); LpEv = orc.fill(3 * n, 0x5EED0F04).reshape(n, 3)
    sec = orc.fill((n << eb) * w, 0x5EED0F05).reshape(n << eb, w)
    ents = [(0, 1, False), (3, 1, True), (4, 3, False), (6, 1, True)]
    part = orc.evmap(LEv[r0:r1].copy(), LpEv[r0:r1].copy(), r1 - r0, eb, [(sec[r0 << eb: r1 << eb].copy(), c, d, pr) for c, d, pr in ents])
    ev = shard.sharded_evmap(to_t(part))
    q.put((rank, r0, r1, as_u(z).reshape(-1, 3).copy(), ev))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_row_sharded_z_and_evmap_equal_single_process(world, orc):
    n = 96 * world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + world * 11
    procs = [ctx.Process(target=_z_ev_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    num = orc.fill(3 * n, 0x5EED0F01).reshape(n, 3); den = orc.fill(3 * n, 0x5EED0F02).reshape(n, 3)
    z = orc.calculate_z(num, den)
    eb, w = 1, 7
    LEv = orc.fill(3 * n, 0x5EED0F03).reshape(n, 3); LpEv = orc.fill(3 * n, 0x5EED0F04).reshape(n, 3)
    sec = orc.fill((n << eb) * w, 0x5EED0F05).reshape(n << eb, w)
    ev = orc.evmap(LEv, LpEv, n, eb, [(sec, c, d, pr) for c, d, pr in [(0, 1, False), (3, 1, True), (4, 3, False), (6, 1, True)]])
    for rank, r0, r1, zl, evl in res:
        assert (zl == z[r0:r1]).all(), rank
        assert evl == [[int(v) for v in e] for e in ev], rank


def _h1h2_worker(rank, world, port, n, dim, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import gl_oracle as orc
    from xlayer_prover_amd import shard
    as_u = lambda t: t.numpy().view(np.uint64)
    to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1))
    base = orc.fill(40 * dim, 3).reshape(40, dim)
    rng = np.random.default_rng(5)
    t = base[np.sort(rng.integers(0, 40, n))]; f = t[rng.integers(0, n, n)]
    r0, r1 = shard.row_range(n, world, rank)

    def h1h2(ff, tt):
        a, b = orc.calculate_h1h2(as_u(ff), as_u(tt), dim)
        return to_t(a), to_t(b)
    h1, h2 = shard.sharded_h1h2(to_t(f[r0:r1]), to_t(t[r0:r1]), dim, h1h2)
    q.put((rank, r0, r1, as_u(h1).copy(), as_u(h2).copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dim", [(2, 1), (4, 3)])
def test_row_sharded_h1h2_equals_single_process(world, dim, orc):
    n = 64 * world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000) + world * 17 + dim
    procs = [ctx.Process(target=_h1h2_worker, args=(r, world, port, n, dim, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = orc.fill(40 * dim, 3).reshape(40, dim)
    rng = np.random.default_rng(5)
    t = base[np.sort(rng.integers(0, 40, n))]; f = t[rng.integers(0, n, n)]
    h1, h2 = orc.calculate_h1h2(f, t, dim)
    h1, h2 = np.asarray(h1).reshape(n, dim), np.asarray(h2).reshape(n, dim)
    for rank, r0, r1, a, b in res:
        assert (a.reshape(-1, dim) == h1[r0:r1]).all() and (b.reshape(-1, dim) == h2[r0:r1]).all(), rank


def _genproof_worker(rank, world, port, nbits, fri_steps, nq, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import gl_oracle as orc
    from tests import mini_backends as mb, mini_genproof_sharded as sh
    out = sh.prove_sharded(mb.OracleBackend(orc), nbits, fri_steps, nq, seed=seed)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_genproof_equals_single_process_proof(world, orc):
    """the whole toy genProof (commit, lookup, grand product, quotient, evaluations, FRI, queries) row-sharded over `world`
    ranks yields the single-process proof bit for bit, on every rank, and the independent verifier accepts it"""
    from tests import mini_backends as mb, mini_genproof_flow as flow, mini_verifier
    nbits, fri_steps, nq, seed = 7, [8, 5, 3], 8, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000) + world * 13
    procs = [ctx.Process(target=_genproof_worker, args=(r, world, port, nbits, fri_steps, nq, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = flow.prove(mb.OracleBackend(orc), nbits, fri_steps, nq, seed=seed)
    owners = {y // ((1 << (nbits + 1)) // world) for y in exp["fri"]["queries"]}
    assert len(owners) > 1                                             # queries really are opened by different ranks
    for rank, out in res:
        for k in exp:
            assert out[k] == exp[k], (rank, k)
    mini_verifier.verify(orc, res[0][1], nbits, fri_steps, nq, mb.gl_w)
