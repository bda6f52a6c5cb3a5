"""The ROW-SHARDED genProof at the zkEVM geometry (tools/zkevm_genproof_sharded.py: windows + halos for the step programs incl. their
next-row stores, rows <-> columns exchanges around the LDE, sub-trees + cap tree, sharded Z / evmap, replicated H1H2 / quotient /
FRI, openings by the owning rank) must print the one-GPU run's proof (tools/bench_zkevm_genproof.py, which tests/test_full_size.py
checks against the oracle): in a world of one, through RCCL in a world of one, and as 2 and 4 PROCESSES sharing the test box's GPU
(gloo, device tensors staged through the host -- RCCL wants one GPU per rank)."""
import argparse
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LN = 12


def _built():
    from xlayer_prover_amd import programs as R
    return os.path.exists(os.path.join(R.REF_DIR, "zkevm.step52ns.xlpexpr"))


def _reference(ln):
    from tools import bench_zkevm_genproof as one
    a = argparse.Namespace(log_n=ln, circuit="zkevm", batch=False, fri_steps="", queries=0, lookups=-1, zs=-1, no_jit=True, no_dots=False, vm_only="",
                           no_opt=False, aliased=0)
    out = one.prove(a, "zkevm", ln)
    return out["roots"], out["fri_roots_and_head_of_final_pol"]


def _check_openings(orc, res, ne_total):
    """sampled step-0 openings of the sharded proof verify against its roots on the oracle"""
    names = ["cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns"]
    ys = res["queries"]
    for qi in (0, 1, len(ys) // 2, len(ys) - 1):
        for k, name in enumerate(names):
            row, sib = res["openings"][(qi, name)]
            assert len(row) == res["widths"][name]
            r = orc.merkle_root_from_path(row, sib, ys[qi])
            assert [int(x) for x in r] == res["roots"][k], (qi, name)


@pytest.mark.gpu
@pytest.mark.parametrize("coll", [True])
def test_sharded_zkevm_genproof_world_of_one(orc, coll):
    import torch
    import torch.distributed as dist
    from xlayer_prover_amd import capi
    from tools import zkevm_genproof_sharded as sh
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if not _built():
        pytest.skip("converted reference programs not built")
    capi.init(0); capi.use_torch_stream()
    exp_roots, exp_fri = _reference(LN)
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29641")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got = sh.prove_sharded(LN, collectives=coll)
    finally:
        if coll:
            dist.destroy_process_group()
    assert got["roots"] == exp_roots
    assert got["fri_roots_and_head_of_final_pol"] == exp_fri
    _check_openings(orc, got, 1 << (LN + 1))


def _worker(rank, world, port, ln, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from xlayer_prover_amd import capi
    from tools import zkevm_genproof_sharded as sh
    torch.cuda.set_device(0)
    capi.init(0); capi.use_torch_stream()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = sh.prove_sharded(ln)
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2])
def test_sharded_zkevm_genproof_ranks_on_one_gpu(orc, world):
    import multiprocessing as mp
    import queue
    import time
    import torch
    from xlayer_prover_amd import capi
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if not _built():
        pytest.skip("converted reference programs not built")
    capi.init(0); capi.use_torch_stream()
    exp_roots, exp_fri = _reference(LN)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36700 + (os.getpid() % 1500) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, LN, q)) for r in range(world)]
    for p in procs:
        p.start()
    res, t0 = [], time.time()
    while len(res) < world:
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 400:
                for p in procs:
                    p.kill()
                pytest.fail(f"rank processes failed: exit codes {dead}")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    owners = {y // ((1 << (LN + 1)) // world) for y in res[0][1]["queries"]}
    assert len(owners) > 1                                    # the openings really came from different ranks
    for rank, out in res:
        assert out["roots"] == exp_roots, rank
        assert out["fri_roots_and_head_of_final_pol"] == exp_fri, rank
        _check_openings(orc, out, 1 << (LN + 1))


@pytest.mark.gpu
def test_rccl_exchange_of_a_large_section_world_of_one():
    """A section of 751 x 2^18 words (1.5 GiB) through the RCCL exchange paths of shard.py: both directions must be the identity in a
    world of one.  (As ONE all_to_all_single message this size comes back with its second half unwritten from the RCCL of this image;
    shard._exchange therefore moves sections in rounds of <= 256 MiB per message and copies the own block locally.)"""
    import torch
    import torch.distributed as dist
    from xlayer_prover_amd import capi, shard
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.init(0); capi.use_torch_stream()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29643")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    old = shard.FORCE_EXCHANGE
    try:
        shard.FORCE_EXCHANGE = True
        n, w = 1 << 18, 751
        x = torch.empty(n * w, dtype=torch.int64, device="cuda"); capi.dev_fill(x, 0x5EED0777, 0)
        cols = shard.rows_to_columns(x.view(n, w), n)
        assert bool((cols.reshape(-1) == x).all())
        rows = shard.columns_to_rows(cols, w)
        assert bool((rows.reshape(-1) == x).all())
    finally:
        shard.FORCE_EXCHANGE = old
        dist.destroy_process_group()
