#!/usr/bin/env python3
"""Section commit across G GPUs (xlayer_prover_amd/shard.py::sharded_commit): column-shard LDE -> one all-to-all
over RCCL/xGMI -> row-shard Merkle sub-tree -> 32-byte-per-rank cap all-gather -> root on every rank.  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 tools/bench_sharded_commit.py \\
        [--log-n 23] [--ncols 751] [--steps 3] [--check]

Prints one JSON line on rank 0: per-stage times (max over ranks), exchanged bytes per rank and the achieved all-to-all rate.
--check (small sizes): the root must equal the single-GPU root of the same trace (computed on rank 0).
--backend gloo runs the ranks on ONE GPU with the exchanges staged through the host (plumbing check only)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from xlayer_prover_amd import capi, shard


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--ncols", type=int, default=751)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) if a.backend == "nccl" else 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    capi.init(local); capi.use_torch_stream()
    n, n_ext, W = 1 << a.log_n, 2 << a.log_n, a.ncols
    c0, c1 = shard.col_range(W, world, rank)
    wc = c1 - c0
    # rank's columns of the synthetic trace: element (row, col) = fill(seed, row * W + col)
    full_seed = 0x5EED0003
    if a.check:
        full = torch.empty(n * W, dtype=torch.int64, device=dev); capi.dev_fill(full, full_seed, 0)
        trace = full.view(n, W)[:, c0:c1].contiguous()
        del full                                               # (rank 0 fills it again for the check: 50 GB at the zkEVM size)
        torch.cuda.empty_cache()
    else:
        trace = torch.empty(n * wc, dtype=torch.int64, device=dev); capi.dev_fill(trace, full_seed + rank, 0); trace = trace.view(n, wc)
    t = {"lde": 0.0, "all_to_all": 0.0, "merkle": 0.0, "caps": 0.0}

    def sync():
        torch.cuda.synchronize()

    def once(record):
        sync(); dist.barrier(); sync(); t0 = time.perf_counter()
        ext = torch.empty((n_ext, wc), dtype=torch.int64, device=dev)
        capi.dev_extend(ext.view(-1), trace.view(-1), n_ext, n, wc)
        sync(); t1 = time.perf_counter()
        rows = shard.columns_to_rows(ext, W)
        sync(); t2 = time.perf_counter()
        h = rows.shape[0]
        nodes = torch.empty((2 * h - 1) * 4, dtype=torch.int64, device=dev)
        capi.dev_merkle(nodes, rows.view(-1), W, h)
        sync(); t3 = time.perf_counter()
        root, _ = shard.assemble_root(nodes[-4:].clone(), shard.gpu_hash_pairs)
        sync(); t4 = time.perf_counter()
        if record:
            for k, v in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                t[k] += v
        return root

    once(False)
    for _ in range(a.steps):
        root = once(True)
    tt = torch.tensor([v / a.steps for v in t.values()], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ok = None
    del trace
    torch.cuda.empty_cache()
    if a.check and rank == 0:
        full = torch.empty(n * W, dtype=torch.int64, device=dev); capi.dev_fill(full, full_seed, 0)
        ext = torch.empty(n_ext * W, dtype=torch.int64, device=dev)
        nodes = torch.empty((2 * n_ext - 1) * 4, dtype=torch.int64, device=dev)
        capi.dev_commit(ext, nodes, full, n_ext, n, W)
        torch.cuda.synchronize()
        ok = bool((nodes[-4:] == root.to(dev)).all().item())
        del full, ext, nodes
    if rank == 0:
        lde, a2a, mk, caps = [float(x) for x in tt.tolist()]
        sent = 8 * n_ext * wc * (world - 1) / world
        print(json.dumps({"workload": f"sharded section commit 2^{a.log_n} x {W} -> 2^{a.log_n + 1}, {world} ranks, backend {a.backend}",
                          "n_gpus": world, "stage_s": {"lde": round(lde, 5), "exchange_and_unpack": round(a2a, 5), "merkle_subtree": round(mk, 5), "caps_all_gather_and_top": round(caps, 5)},
                          "commit_s": round(lde + a2a + mk + caps, 5), "bytes_sent_per_rank": int(sent),
                          "exchange_GBps_sent_per_rank": round(sent / a2a / 1e9, 1) if a2a > 0 and world > 1 else None,
                          "root": [int(v) & 0xFFFFFFFFFFFFFFFF for v in root.tolist()], "root_equals_single_gpu_commit": ok}), flush=True)
        if a.check:
            assert ok, "sharded root differs from the single-GPU root"
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
