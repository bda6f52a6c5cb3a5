#!/usr/bin/env python3
"""bench.py -- headline benchmark of the STARK hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input:
BASELINE.json configs[1], the LDE (NTT_Goldilocks::extendPol, starks.cpp:53)
of a 2^23 x 8 Goldilocks trace to 2^24 x 8 (blow-up 2), inputs resident in HBM.
Metric: Goldilocks NTT throughput in GFE/s = output field elements / second,
whole job (all ranks).  With --gpus N each rank extends its own 8-column shard
of an (8N)-column trace (columns are independent: weak scaling, no data-path
collective).

Prints ONE JSON line (rank 0) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 23
NCOLS = 8
EXT_BITS = 1
SEED = 0x5EED0002
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LDE_KERNELS = ["xlp::lde_top_dif<11>", "xlp::lde_mid", "xlp::lde_top_dit<11>"]   # one launch of each per extendPol
VALU_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1", "lde_valu.json")
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r1", "lde_traffic.json")          # PMC passes of tools/traffic_probe.py


def cpu_baseline(log_n, ncols, budget_s=20.0):
    """Oracle (restated CPU path, OpenMP over columns) timed on this host, bounded sample."""
    from oracle import gl_oracle as orc
    threads = orc.num_threads()
    # pick a sample size whose run time is ~10-30 s: calibrate on 2^16
    n = 1 << 16
    src = orc.fill(n * ncols, SEED).reshape(n, ncols)
    t0 = time.perf_counter()
    orc.extend_pol(src, n << EXT_BITS, n, ncols)
    t = time.perf_counter() - t0
    par = min(threads, ncols)
    per_fe_log = t / (n * 17) * (1 if par >= ncols else ncols / par)   # ~ n log n per column-batch model
    lg = 16
    while lg < log_n and per_fe_log * (1 << (lg + 1)) * (lg + 2) < budget_s:
        lg += 1
    n = 1 << lg
    src = orc.fill(n * ncols, SEED).reshape(n, ncols)
    t0 = time.perf_counter()
    orc.extend_pol(src, n << EXT_BITS, n, ncols)
    t = time.perf_counter() - t0
    return {
        "value": (n << EXT_BITS) * ncols / t / 1e9, "unit": "GFE/s", "cores": min(threads, ncols), "kind": "port",
        "sample": f"oracle extendPol 2^{lg} x {ncols} -> 2^{lg + EXT_BITS} x {ncols} ({t:.2f} s, OpenMP over columns, "
                  f"scalar u128 arithmetic; restated CPU path, not the reference's AVX2 binary)",
    }


def genproof_leg(timeout_s=420):
    """The other half of BASELINE.json's metric: wall-clock of the three STARKs of Prover::genBatchProof (prover.cpp:572-645:
    zkEVM 2^23 rows, c12a 2^20, recursive1 2^17), each a whole Starks::genProof (starks.cpp:9-404, every stage, the reference's
    converted step programs) GPU-resident on this one GPU, synthetic witnesses -- tools/bench_zkevm_genproof.py --batch run as a
    child process (the zkEVM proof needs ~293 of the 309 GB).  Reported next to the headline value, never as it; any failure
    is reported as such and does not touch the main line."""
    import subprocess
    import torch
    free_b, total_b = torch.cuda.mem_get_info()
    log_n = 23 if free_b > 296e9 else 22
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_zkevm_genproof.py"), "--batch", "--log-n", str(log_n)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-300:]}
        b = json.loads(r.stdout.strip().splitlines()[-1])
        j = b["proofs"]["STARK_PROOF_BATCH_PROOF"]
    except Exception as e:                                   # missing converted programs, timeout, ...
        return {"error": repr(e)[:300]}
    t = j["timers_s"]
    grp = lambda key: round(sum(v for k, v in t.items() if key(k)), 4)
    return {
        "metric": "genBatchProof STARK wall-clock (s): the three genProof calls of Prover::genBatchProof, one GPU",
        "wall_s": b["genbatchproof_starks_wall_s"], "per_stark_s": b["wall_s"],
        "zkevm": {"log_n": log_n, "wall_s": j["genproof_wall_s"], "sum_timed_s": j["sum_timed_s"],
                  "stages_s": {"merkle": grp(lambda k: "MERKLETREE" in k), "lde": grp(lambda k: k.endswith("_LDE")),
                               "step_programs": grp(lambda k: "CALCULATE_EXPS" in k and "INTT" not in k),
                               "h1h2_z": grp(lambda k: "H1H2" in k or k.endswith("_Z")),
                               "evmap_lev_xdiv": grp(lambda k: "STEP_5" in k and "EXPS" not in k),
                               "q_intt_ntt": grp(lambda k: "INTT" in k), "fri_and_openings": grp(lambda k: k.endswith("_FRI"))},
                  "peak_GB": j.get("peak_GB_torch"), "layout": j.get("layout"), "programs": j["programs"], "workload": j["workload"]},
        "note": "synthetic witnesses; executor and circom witness computation between the STARKs are CPU work outside the hot path",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-genproof", action="store_true", help="skip the whole-genProof leg (N=1 only; ~25 s)")
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--ncols", type=int, default=NCOLS)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from xlayer_prover_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    capi.init(local_rank)
    capi.use_torch_stream()

    n, ncols = 1 << args.log_n, args.ncols
    n_ext = n << EXT_BITS
    src = torch.empty(n * ncols, dtype=torch.int64, device=dev)
    out = torch.empty(n_ext * ncols, dtype=torch.int64, device=dev)
    scratch = torch.empty(2 * n * ncols, dtype=torch.int64, device=dev)
    # rank r owns columns [8r, 8r+8) of the synthetic trace: element index = row*(8*world) + col
    capi.dev_fill(src, SEED, rank * n * ncols)

    def step():
        capi.dev_extend(out, src, n_ext, n, ncols, scratch)

    for _ in range(args.warmup):
        step()
    # per-kernel HIP events (recorded inside the library on the stream the kernels run on)
    torch.cuda.synchronize()
    capi.profile_enable(True)
    for _ in range(3):
        step()
    kern_ms_sum, kern_cnt = capi.profile_read()
    capi.profile_enable(False)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    dev_ms = ev0.elapsed_time(ev1)           # HIP events on the stream the kernels run on
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        fe_per_step = n_ext * ncols * world
        value = fe_per_step / (elapsed / args.steps) / 1e9
        # roofline.  Algorithmic bytes of one LDE = 8*N*ncols in + 8*N_ext*ncols out (24 B
        # per input element at blow-up 2).  One LDE = one launch each of lde_top_dif, lde_mid,
        # lde_top_dit; the dominant kernel is lde_mid.  `achieved` prices the whole LDE: algorithmic
        # bytes / (sum of the three kernels' average launch durations, HIP events on their stream).
        alg_bytes_lde = 8 * n * ncols + 8 * n_ext * ncols
        per_kernel_ms = [x / kern_cnt for x in kern_ms_sum] if kern_cnt else [None] * 3
        lde_kernel_ms = sum(per_kernel_ms) if kern_cnt else dev_ms / args.steps
        achieved = alg_bytes_lde / (lde_kernel_ms * 1e-3) / 1e9
        traffic = None
        if os.path.exists(TRAFFIC_FILE) and args.log_n == LOG_N and ncols == NCOLS:
            traffic = json.load(open(TRAFFIC_FILE))["hbm_bytes_per_lde"]
        # the resource that actually binds: VALU issue slots (measured instruction count of one LDE / measured kernel time)
        valu_issue = None
        if os.path.exists(VALU_FILE) and args.log_n == LOG_N and ncols == NCOLS:
            vj = json.load(open(VALU_FILE))
            rate = vj["valu_wave_instructions_per_lde"] / (lde_kernel_ms * 1e-3)
            valu_issue = {"achieved": round(rate / 1e9, 1), "peak": round(vj["issue_peak_wave_instructions_per_s"] / 1e9, 1), "unit": "G wave-instr/s",
                          "frac": round(rate / vj["issue_peak_wave_instructions_per_s"], 3)}
        roofline = {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "kernel": "xlp::lde_mid (dominant) + lde_top_dif + lde_top_dit = one extendPol",
            "launches_per_step": 3,
            "avg_launch_ms": {k: round(v, 4) for k, v in zip(LDE_KERNELS, per_kernel_ms)} if kern_cnt else None,
            "alg_bytes_per_step": alg_bytes_lde,
            "note": "integer-VALU co-bound (Goldilocks mul = 4 v_mad_u64_u32 + carry chains): see DESIGN.md",
            "valu_issue": valu_issue,
        }
        line = {
            "metric": "Goldilocks NTT GFE/s (LDE 2^%d x %d, blow-up 2)" % (args.log_n, ncols),
            "value": round(value, 3), "unit": "GFE/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64 (Goldilocks p=2^64-2^32+1)", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: extendPol 2^%d x %d -> 2^%d x %d per GPU, row-major, "
                                   "inputs resident in HBM" % (args.log_n, ncols, args.log_n + EXT_BITS, ncols),
                       "columns_total": ncols * world, "sharding": "independent columns per rank, no collective"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_genproof and args.log_n == LOG_N:
            del src, out, scratch
            torch.cuda.empty_cache()
            line["genproof"] = genproof_leg()
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.log_n, ncols)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
