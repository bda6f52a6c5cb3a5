"""The toy prover of tests/mini_genproof_flow.py (every stage of Starks::genProof, starks.cpp:9-404) run on the HIP
library and on the CPU oracle: all roots, evaluations, FRI layers, query indices and opened rows must agree."""
import numpy as np
import pytest

from tests import mini_genproof_flow as flow

from tests.mini_backends import GpuBackend, OracleBackend, gl_w


def test_mini_pil_is_satisfied_on_the_oracle(orc):
    """sanity of the toy system and of the H1H2 / Z restatements together: on the trace domain b = a^2, z[0] = 1 and
    z' den = z num on EVERY row including the wrap-around -- i.e. the grand product over the plookup closes to 1,
    which only happens if h1, h2 are the sorted merge the protocol asks for"""
    out = flow.prove(OracleBackend(orc), 5, [6, 3], 4, seed=3, debug=True)
    assert len(out["evals"]) == 14 and len(out["fri"]["queries"]) == 4
    pols, L = out.pop("_state")
    N = L.N
    def sec(name):
        o, w, rows = L.off[name]
        return pols[o:o + w * rows].reshape(rows, w)
    cm1, z, tmp = sec("cm1_n"), sec("cm3_n"), sec("tmpExp_n")
    assert all(int(cm1[i, 1]) == orc.mul(int(cm1[i, 0]), int(cm1[i, 0])) for i in range(N))
    assert [int(v) for v in z[0]] == [1, 0, 0]
    for i in range(N):
        lhs = orc.mul3(z[(i + 1) % N], tmp[i, 9:12]); rhs = orc.mul3(z[i], tmp[i, 6:9])
        assert (lhs == rhs).all(), i
    # and the proof verifies: transcript, Merkle paths, C(xi) = Q(xi) Zh(xi), FRI polynomial at the queries, folds
    from tests.mini_verifier import verify
    assert verify(orc, out, 5, [6, 3], 4, gl_w)
    bad = dict(out); bad["evals"] = [list(e) for e in out["evals"]]; bad["evals"][4][0] ^= 1
    with pytest.raises(AssertionError):
        verify(orc, bad, 5, [6, 3], 4, gl_w)


@pytest.mark.gpu
@pytest.mark.parametrize("nbits,steps,nq", [(5, [6, 3], 4), (8, [9, 6, 3], 8), (11, [12, 8, 4], 16)])
def test_mini_genproof_gpu_equals_oracle(orc, nbits, steps, nq):
    import torch
    from xlayer_prover_amd import capi
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.init(0); capi.use_torch_stream()
    got = flow.prove(GpuBackend(capi), nbits, steps, nq, seed=nbits)
    exp = flow.prove(OracleBackend(orc), nbits, steps, nq, seed=nbits)
    assert got.keys() == exp.keys()
    for k in exp:
        assert got[k] == exp[k], k
    from tests.mini_verifier import verify
    assert verify(orc, got, nbits, steps, nq, gl_w)          # and what the GPU produced is a valid proof


@pytest.mark.gpu
@pytest.mark.parametrize("nbits,steps,nq,coll", [(6, [7, 4], 5, False), (10, [11, 7, 4], 12, True)])
def test_sharded_genproof_on_one_gpu_equals_oracle(orc, nbits, steps, nq, coll):
    """the multi-GPU form of the flow (tests/mini_genproof_sharded.py: windows with halo rows, rows<->columns exchanges,
    cap tree, partial evmap ...) on one GPU; with coll=True the exchanges go through RCCL in a world of one rank.
    The world-2 / world-4 runs of the same code are in tests/test_shard_gloo.py (CPU, gloo)."""
    import os
    import torch
    import torch.distributed as dist
    from xlayer_prover_amd import capi
    from tests import mini_genproof_sharded as sh
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.init(0); capi.use_torch_stream()
    exp = flow.prove(OracleBackend(orc), nbits, steps, nq, seed=nbits)
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29637")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got = sh.prove_sharded(GpuBackend(capi), nbits, steps, nq, seed=nbits, collectives=coll)
    finally:
        if coll:
            dist.destroy_process_group()
    for k in exp:
        assert got[k] == exp[k], k


def _two_rank_gpu_worker(rank, world, port, nbits, steps, nq, q):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from xlayer_prover_amd import capi
    from tests import mini_genproof_sharded as sh
    from tests.mini_backends import GpuBackend
    torch.cuda.set_device(0)
    capi.init(0); capi.use_torch_stream()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = sh.prove_sharded(GpuBackend(capi), nbits, steps, nq, seed=nbits)
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_genproof_ranks_on_one_gpu_equal_oracle(orc, world):
    """the row-sharded genProof with EVERY kernel on the HIP library and a real exchange between `world` processes: the ranks
    share the one GPU of the test box, so the transport is gloo with the device tensors staged through the host
    (shard._host_staged) -- RCCL wants one GPU per rank; its world-of-one run is the test above.  Every rank's proof equals the
    single-process oracle proof bit for bit."""
    import multiprocessing as mp
    import os
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    nbits, steps, nq = 9, [10, 6, 3], 9
    exp = flow.prove(OracleBackend(orc), nbits, steps, nq, seed=nbits)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36100 + (os.getpid() % 1500) + world
    procs = [ctx.Process(target=_two_rank_gpu_worker, args=(r, world, port, nbits, steps, nq, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    res, t0 = [], time.time()
    while len(res) < world:
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 240:
                for p in procs:
                    p.kill()
                pytest.fail(f"rank processes failed: exit codes {dead}")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    owners = {y // ((1 << (nbits + 1)) // world) for y in exp["fri"]["queries"]}
    assert len(owners) > 1
    for rank, out in res:
        for k in exp:
            assert out[k] == exp[k], (rank, k)


@pytest.mark.gpu
@pytest.mark.parametrize("nbits,steps,nq,optimise", [(6, [7, 4], 5, False), (10, [11, 7, 4], 12, False), (9, [10, 6, 3], 9, True)])
def test_cpp_starks_genproof_equals_flow_and_verifies(orc, tmp_path, nbits, steps, nq, optimise):
    """The C++ GPU-resident `Starks::genProof` mirror (host/starks.hpp: StarkInfo-driven, starks.cpp:9-404 stage for stage)
    on the mini PIL: its zkin proof equals what the Python flow computes on the oracle, and passes the verifier.
    optimise: the step programs run in their dot-product form (tools/optimize_programs.py pairs picked up by XlpSteps)."""
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xlayer_prover_amd", "host", "test_host_shim")
    flow.toy_stark_files(str(tmp_path), "toy", nbits, steps, nq, seed=nbits, optimise=optimise)
    r = subprocess.run([exe, "genproof", str(tmp_path), "toy"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert ("dot-product form" in r.stderr) == optimise
    Z = json.loads(r.stdout)
    exp = flow.prove(OracleBackend(orc), nbits, steps, nq, seed=nbits)
    ints = lambda v: [int(x) for x in v]
    for k, name in (("root1", "root1"), ("root2", "root2"), ("root3", "root3"), ("root4", "root4"), ("verkey", "verkey"), ("publics", "publics")):
        assert ints(Z[name]) == exp[k], name
    assert [ints(e) for e in Z["evals"]] == exp["evals"]
    assert [x for e in Z["finalPol"] for x in ints(e)] == exp["fri"]["finalPol"]
    for si in range(1, len(steps)):
        assert ints(Z[f"s{si}_root"]) == exp["fri"]["roots"][si - 1]
        for q in range(nq):
            row, sib = exp["fri"]["si"][si - 1][q]
            assert ints(Z[f"s{si}_vals"][q]) == row and [ints(s) for s in Z[f"s{si}_siblings"][q]] == sib
    for q in range(nq):
        for k, tname in enumerate(["1", "2", "3", "4", "C"]):
            row, sib = exp["fri"]["s0"][q][k]
            assert ints(Z[f"s0_vals{tname}"][q]) == row and [ints(s) for s in Z[f"s0_siblings{tname}"][q]] == sib
    # the C++ proof as a verifier input
    from tests.mini_verifier import verify
    got = dict(exp)
    got["evals"] = [ints(e) for e in Z["evals"]]; got["root1"] = ints(Z["root1"]); got["root4"] = ints(Z["root4"])
    assert verify(orc, got, nbits, steps, nq, gl_w)
