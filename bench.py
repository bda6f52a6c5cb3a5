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
LDE_KERNELS = ["xlp::lde_top_dif<11>", "xlp::lde_mid_mfma<false>", "xlp::lde_top_dit<11>"]   # one launch of each per extendPol (the middle pass: the matrix-pipe kernel, round 5)


def cpu_baseline(log_n, ncols, budget_s=20.0):
    """Oracle (restated CPU path) timed on this host, bounded sample, EVERY host core: orc_extend_pol_team puts the whole OpenMP team on all
    columns together (the first 16 stages of a transform inside 2^16-point blocks, the rest stage by stage: butterflies of a stage are
    independent), so an 8-column trace uses all 128 cores of the GPU box."""
    from oracle import gl_oracle as orc
    threads = orc.num_threads()
    # pick a sample size whose run time is ~10-30 s: calibrate on 2^16
    n = 1 << 16
    src = orc.fill(n * ncols, SEED).reshape(n, ncols)
    t0 = time.perf_counter()
    orc.extend_pol_team(src, n << EXT_BITS, n, ncols)
    t = time.perf_counter() - t0
    per_fe_log = t / (n * 17)                                  # ~ n log n model (small sizes parallelise worse: the estimate is an upper bound)
    lg = 16
    while lg < log_n and per_fe_log * (1 << (lg + 1)) * (lg + 2) < budget_s:
        lg += 1
    n = 1 << lg
    src = orc.fill(n * ncols, SEED).reshape(n, ncols)
    t0 = time.perf_counter()
    orc.extend_pol_team(src, n << EXT_BITS, n, ncols)
    t = time.perf_counter() - t0
    out = {
        "value": (n << EXT_BITS) * ncols / t / 1e9, "unit": "GFE/s", "cores": threads, "kind": "port",
        "sample": f"oracle extendPol 2^{lg} x {ncols} -> 2^{lg + EXT_BITS} x {ncols} ({t:.2f} s, OpenMP team of {threads} threads on all columns together, blocked stages, "
                  f"scalar u128 arithmetic; restated CPU path, not the reference's AVX2 binary)",
    }
    # the other two legs of the path on the same host cores (bounded samples): the Merkle leaf sponge + tree, and one small complete genProof
    legs = {}
    try:
        h, w = 1 << 14, 751
        rows = orc.fill(h * w, SEED + 1).reshape(h, w)
        t0 = time.perf_counter()
        orc.merkletree(rows, w, h)
        tm = time.perf_counter() - t0
        perms = h * ((w + 7) // 8) + h - 1
        legs["merkle"] = {"value": round(perms / tm, 1), "unit": "Poseidon permutations/s", "cores": threads, "kind": "port",
                          "sample": f"oracle merkletree 2^14 x 751 ({perms} permutations, {tm:.2f} s, OpenMP over rows); the GPU figure beside it is roofline_merkle.permutations_per_s"}
    except Exception as e:
        legs["merkle"] = {"error": repr(e)[:200]}
    try:
        from tests import mini_genproof_flow as flow
        from tests.mini_backends import OracleBackend
        nb = 14
        steps = [nb + 1, 10, 5]
        t0 = time.perf_counter()
        res = flow.prove(OracleBackend(orc), nb, steps, 16, seed=7)
        tg = time.perf_counter() - t0
        legs["genproof_toy"] = {"value": round(tg, 3), "unit": "s", "cores": threads, "kind": "port", "root1": res["root1"],
                                "sample": f"the complete stage sequence of Starks::genProof on the toy PIL (tests/mini_genproof_flow.py: 17 committed columns, N = 2^{nb}, "
                                          f"FRI {steps}, 16 queries) on the oracle; the same flow on the GPU: genproof_toy_gpu_s"}
    except Exception as e:
        legs["genproof_toy"] = {"error": repr(e)[:200]}
    out["legs"] = legs
    return out


def genproof_toy_gpu(capi):
    """the toy-PIL genProof of cpu_baseline.legs.genproof_toy on the GPU (same sizes), for the ratio only"""
    try:
        from tests import mini_genproof_flow as flow
        from tests.mini_backends import GpuBackend
        capi.init(0); capi.use_torch_stream()
        B = GpuBackend(capi)
        flow.prove(B, 10, [11, 6], 4)
        t0 = time.perf_counter()
        res = flow.prove(B, 14, [15, 10, 5], 16, seed=7)
        return {"s": round(time.perf_counter() - t0, 4), "root1": res["root1"]}
    except Exception as e:
        return {"error": repr(e)[:200]}


def counters():
    """newest profiles/r*/counters.json (tools/collect_counters.py: rocprofv3 --pmc passes), with a staleness verdict: the file is
    stamped with the hash of the kernel sources it was measured on"""
    import glob
    from xlayer_prover_amd.srchash import csrc_hash
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "counters.json")), key=lambda f: int(os.path.basename(os.path.dirname(f))[1:]))
    if not files:
        return None, True, None
    c = json.load(open(files[-1]))
    return c, c.get("csrc_sha256") != csrc_hash(), os.path.relpath(files[-1], ROOT)


def lde_colmajor_leg(capi, torch, dev, log_n):
    """What genBatchProof runs since round 4: extendPol on COLUMN-MAJOR sections (csrc/ntt_lde_cm.hip, ntt_lde_cm_mid.hip: lde_cm_top_dif, lde_cm_mid,
    lde_cm_top_dit), at the headline's shape (x 8) and at a proof's chunk width (x 64): GFE/s of the whole call, the three kernels' average launch
    durations (HIP events inside the library), and the counter figures of the newest profiles/rN/counters.json ("lde_colmajor")."""
    n, ne = 1 << log_n, 2 << log_n
    out = {"kernel": "xlp::lde_cm_top_dif + lde_cm_mid + lde_cm_top_dit = one extendPol on column-major sections (the product path of genBatchProof)", "bound": "valu"}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for w in (8, 64):
        src = torch.empty(n * w, dtype=torch.int64, device=dev); capi.dev_fill(src, SEED, 0)
        dst = torch.empty(ne * w, dtype=torch.int64, device=dev)
        fn = lambda: capi.dev_extend_cm(dst, ne, src, n, ne, n, w)
        fn(); fn(); torch.cuda.synchronize()
        reps = 5
        capi.profile_enable(True)
        for _ in range(reps):
            fn()
        ms, cnt = capi.profile_read()
        capi.profile_enable(False)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        tot = e0.elapsed_time(e1) / reps
        kms = sum(ms) / reps
        alg = 8 * n * w + 8 * ne * w
        out["x%d" % w] = {"GFE_s": round(ne * w / tot / 1e6, 2), "ms_per_lde": round(tot, 4), "kernel_ms_per_lde": [round(x / reps, 4) for x in ms], "chunks": cnt // reps,
                          "hbm": {"achieved": round(alg / (kms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "alg_bytes_per_lde": alg}}
        del src, dst
    cj, stale, cfile = counters()
    if cj and cj.get("lde_colmajor"):
        out["counters"] = dict(cj["lde_colmajor"], file=cfile, stale=stale)
    return out


def merkle_leg(capi, torch, dev):
    """The kernel that is the largest share of a proof: the leaf sponge of PoseidonGoldilocks::merkletree (linear_hash_kernel),
    timed live with HIP events on the library's stream on 2^20 x 751 rows; the counter-derived figures come from
    profiles/rN/counters.json measured on the same shape."""
    h, w = 1 << 20, 751
    rows = torch.empty(h * w, dtype=torch.int64, device=dev); nodes = torch.empty((2 * h - 1) * 4, dtype=torch.int64, device=dev)
    capi.dev_fill(rows, 0x5EED0003, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3

    def timed(fn):
        fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms_row = timed(lambda: capi.dev_linear_hash(nodes, rows, w, h))
    # the product path (round 4): the same words read as COLUMNS (the buffer taken as 751 columns of 2^20 words: timing does not depend on the values)
    ms = timed(lambda: capi.dev_linear_hash_absorb_cm(nodes, rows, h, 0, w, h, 0, True, True))
    perms = h * ((w + 7) // 8)
    alg = 8 * h * w + 32 * h
    c, stale, src = counters()
    out = {"kernel": "xlp::linear_hash_absorb_cm_kernel (leaf sponge of merkletree over a column-major section, 2^20 x 751 rows; the product path)", "bound": "valu", "avg_launch_ms": round(ms, 4),
           "row_major_sections": {"kernel": "xlp::linear_hash_kernel", "avg_launch_ms": round(ms_row, 4), "permutations_per_s": round(h * ((w + 7) // 8) / ms_row * 1e3, 0)},
           "permutations_per_s": round(perms / ms * 1e3, 0), "hbm": {"achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4),
                                                                     "alg_bytes_per_launch": alg}}
    # the ceiling SURVEY 8d asks for: the same permutations per lane on the same grid with the state resident in registers (no HBM traffic at all)
    try:
        sink = torch.empty(h, dtype=torch.int64, device=dev)
        it = (w + 7) // 8
        capi._ck(capi.load().xlp_dev_perm_resident(capi._dp(sink), h, it), "xlp_dev_perm_resident")
        e0.record()
        for _ in range(reps):
            capi._ck(capi.load().xlp_dev_perm_resident(capi._dp(sink), h, it), "xlp_dev_perm_resident")
        e1.record(); torch.cuda.synchronize()
        ms_res = e0.elapsed_time(e1) / reps
        out["alu_ceiling"] = {"kernel": "xlp::perm_resident_kernel: the same 94 chained permutations per lane on the same grid, state resident in registers, no memory traffic",
                              "avg_launch_ms": round(ms_res, 4), "permutations_per_s": round(perms / ms_res * 1e3, 0), "unit": "Poseidon permutations/s",
                              "frac": round(ms_res / ms, 4), "note": "frac = sponge rate / this ceiling: what is missing is what the sponge's loads cost"}
    except Exception as e:                                   # noqa: BLE001
        out["alu_ceiling"] = {"error": repr(e)[:200]}
    if c and (c.get("merkle_colmajor") or "merkle" in c):
        m = c.get("merkle_colmajor") or c["merkle"]
        rate = m["valu_wave_instructions_linear_hash"] / (ms * 1e-3)
        out["valu_issue"] = {"achieved": round(rate / 1e9, 1), "unit": "G wave-instr/s", "lane_instructions_per_permutation": round(m["valu_lane_instructions_per_permutation"], 0),
                             "note": "instructions retired per second (counter-derived); the ceiling of this kernel is alu_ceiling above, measured, not a model of issue rates"}
        out["hbm"]["traffic"] = m["hbm_bytes_linear_hash"]
        out["counters"] = src; out["counters_stale"] = stale
    del rows, nodes
    return out


def bn128_leg(capi, torch, dev):
    """SURVEY 8 f4 kernels, timed live (HIP events on the library's stream): MerkleTreeBN128::merkelize at the recursivef geometry
    (2^23 leaves x 12 columns), FFT<RawFr>::fft of 2^24 elements and the G1 multiexp of 2^22 points (the fflonk prover's two
    primitives).  Synthetic inputs; None when the Poseidon-BN128 tables were not generated."""
    import numpy as np
    try:
        capi.load().xlp_bn128_init(None)
        if capi.load().xlp_bn128_init(None) != 0:
            return {"error": capi.load().xlp_last_error().decode()}
    except Exception as e:     # noqa: BLE001
        return {"error": str(e)}

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    out = {}
    h, w = 1 << 23, 12
    src = torch.empty(h * w, dtype=torch.int64, device=dev); capi.dev_fill(src, 0x5EED0F04, 0)
    nodes = torch.empty(capi.merkle_bn128_num_nodes(h) * 4, dtype=torch.int64, device=dev)
    ms = timed(lambda: capi.dev_merkle_bn128(nodes, src, w, h), 2)
    out["merkle_bn128"] = {"workload": "MerkleTreeBN128::merkelize 2^23 x 12 (recursivef cm1)", "ms": round(ms, 2), "leaf_permutations_t5": h, "node_permutations_t17": 559241}
    del src, nodes
    n = 1 << 24
    a = torch.empty(n * 4, dtype=torch.int64, device=dev); capi.dev_fill(a, 0x5EED0F05, 0)
    a.view(n, 4)[:, 3] &= 0x0FFFFFFFFFFFFFFF
    ms = timed(lambda: capi.dev_fft_bn254(a, n), 3)
    out["fft_bn254"] = {"workload": "FFT<RawFr>::fft 2^24", "ms": round(ms, 3), "Melem_per_s": round(n / ms / 1e3, 1)}
    del a
    q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    R = (1 << 256) % q
    pts, cur = [], None                       # 256 multiples of the generator (host integers), replicated
    for _ in range(256):
        if cur is None:
            cur = (1, 2)
        else:
            lam = (3 * cur[0] * cur[0] * pow(2 * cur[1], -1, q) if cur == (1, 2) else (2 - cur[1]) * pow(1 - cur[0], -1, q)) % q
            x = (lam * lam - cur[0] - 1) % q
            cur = (x, (lam * (cur[0] - x) - cur[1]) % q)
        pts.append(cur)
    tab = np.array([[((c * R % q) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for c in p for i in range(4)] for p in pts], dtype=np.uint64)
    n = 1 << 22
    bases = capi.from_np(tab[np.random.default_rng(1).integers(0, 256, size=n)].reshape(-1))
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev); capi.dev_fill(sc, 0x5EED0F06, 0)
    res = torch.empty(8, dtype=torch.int64, device=dev)
    ms = timed(lambda: capi.dev_msm_bn254(res, bases, sc, n), 2)
    out["msm_bn254"] = {"workload": "G1 multiexp 2^22 points, 256-bit scalars", "ms": round(ms, 2), "Mpoints_per_s": round(n / ms / 1e3, 1)}
    # the bucket pass against its ALU ceiling: n x windows mixed additions per multi-exponentiation; the ceiling = the same additions chained
    # on register-resident operands (xlp_dev_msm_resident), one lane per bucket as in the pass, 64 additions per lane
    lib = capi.load()
    c = lib.xlp_msm_bn254_window_bits(n)
    nwin = (256 + c - 1) // c
    lanes, iters = nwin << c, 64
    sink = torch.empty(lanes * 12, dtype=torch.int64, device=dev)
    one_pt = bases[:8].clone()
    ms_r = timed(lambda: capi._ck(lib.xlp_dev_msm_resident(capi._dp(sink), capi._dp(one_pt), lanes, iters), "xlp_dev_msm_resident"), 2)
    ceil_rate = lanes * iters / (ms_r * 1e-3)
    rate = n * nwin / (ms * 1e-3)
    out["msm_bn254"]["roofline"] = {"bound": "valu", "unit": "mixed G1 additions/s", "window_bits": c, "windows": nwin, "additions_per_multiexp": n * nwin,
                                    "achieved": round(rate, 0), "peak": round(ceil_rate, 0), "frac": round(rate / ceil_rate, 4),
                                    "peak_from": "xlp::g1::msm_resident_kernel: %d lanes x %d chained additions of one cached point, %.3f ms (measured in this run; no HBM traffic)" % (lanes, iters, ms_r),
                                    "note": "achieved counts the WHOLE multi-exponentiation (keys, sort, bucket pass, window reductions, Horner) against the bucket pass's ceiling"}
    return out


def sharded_commit_leg(capi, torch, dist, dev, world, rank, log_n=23, ncols=751, check=True):
    """SURVEY.md 8e / BASELINE north_star: ONE zkEVM-width section (2^23 x 751, the committed trace cm1) committed by all ranks
    together -- rank g extends its ~751/G columns, one all-to-all over RCCL/xGMI turns column shards into row shards, each rank
    hashes the sub-tree over its rows, a 32-byte-per-rank all-gather assembles the caps and every rank hashes the top log2 G
    levels.  Strong scaling (the section is fixed).  Rank 0 then commits the same section alone and the roots must agree."""
    import time
    from xlayer_prover_amd import shard
    n, n_ext, W = 1 << log_n, 2 << log_n, ncols
    c0, c1 = shard.col_range(W, world, rank)
    wc = c1 - c0
    seed = 0x5EED0003
    full = torch.empty(n * W, dtype=torch.int64, device=dev)
    capi.dev_fill(full, seed, 0)                              # element (row, col) = fill(seed, row * W + col), the same on every rank
    trace = full.view(n, W)[:, c0:c1].contiguous()
    del full
    torch.cuda.empty_cache()
    t = [0.0, 0.0, 0.0, 0.0]

    def once(record):
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ext = torch.empty((n_ext, wc), dtype=torch.int64, device=dev)
        capi.dev_extend(ext.view(-1), trace.view(-1), n_ext, n, wc)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        rows = shard.columns_to_rows(ext, W)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        del ext
        h = rows.shape[0]
        nodes = torch.empty((2 * h - 1) * 4, dtype=torch.int64, device=dev)
        capi.dev_merkle(nodes, rows.view(-1), W, h)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        root, _ = shard.assemble_root(nodes[-4:].clone(), shard.gpu_hash_pairs)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        if record:
            for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                t[i] += v
        return root
    once(False)
    reps = 2
    for _ in range(reps):
        root = once(True)
    tt = torch.tensor([v / reps for v in t], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    lde, a2a, mk, caps = [float(x) for x in tt.tolist()]
    ok = None
    del trace
    torch.cuda.empty_cache()
    if check and rank == 0:
        full = torch.empty(n * W, dtype=torch.int64, device=dev); capi.dev_fill(full, seed, 0)
        ext = torch.empty(n_ext * W, dtype=torch.int64, device=dev); nodes = torch.empty((2 * n_ext - 1) * 4, dtype=torch.int64, device=dev)
        capi.dev_commit(ext, nodes, full, n_ext, n, W)
        torch.cuda.synchronize()
        ok = bool((nodes[-4:] == root.to(dev)).all().item())
        del full, ext, nodes
        torch.cuda.empty_cache()
    dist.barrier()
    sent = 8 * n_ext * wc * (world - 1) / world
    return {"workload": f"one section 2^{log_n} x {W} -> 2^{log_n + 1} x {W} + Merkle root, columns sharded over {world} ranks (strong scaling)",
            "stage_s": {"lde": round(lde, 5), "exchange_and_unpack": round(a2a, 5), "merkle_subtree": round(mk, 5), "caps_all_gather_and_top": round(caps, 5)},
            "commit_s": round(lde + a2a + mk + caps, 5), "bytes_sent_per_rank": int(sent),
            "exchange_GBps_sent_per_rank": round(sent / a2a / 1e9, 1) if a2a > 0 and world > 1 else None,
            "root": [int(v) & 0xFFFFFFFFFFFFFFFF for v in root.tolist()], "root_equals_single_gpu_commit": ok}


def genproof_leg(timeout_s=300):
    """The other half of BASELINE.json's metric: wall-clock of Prover::genBatchProof's three STARKs (prover.cpp:572-645: zkEVM 2^23
    rows, c12a 2^20, recursive1 2^17), each a whole Starks::genProof (starks.cpp:9-404, every stage, the reference's converted step
    programs), through the C++ entry points with the reference's shape (host/prover.hpp, host/starks.hpp; binary
    host/zkprover_batch), GPU-resident on this one GPU, the 50.4 GB zkEVM witness uploaded from page-locked host memory inside
    the wall.  Run twice: witness upload pipelined with step 1 (the product path) and uploaded in one piece first (what the
    pipeline hides).  Reported next to the headline value, never as it; a failure is reported as such."""
    import subprocess
    exe = os.path.join(ROOT, "xlayer_prover_amd", "host", "zkprover_batch")
    # the device sections are COLUMN-MAJOR (bundles *.geo23ac / geo20c / geo17c; include/xlp_gl.h): the same proof as on the reference's row-major slab,
    # bit for bit (tests/test_cpp_colmajor.py compares the layouts once, at 2^12 / 2^13 rows; no row-major bundle is built since round 5)
    cfg = os.path.join(ROOT, "tools", "config_runFile_BatchProof_synthetic_colmajor.json")
    starks = ("STARK_PROOF_BATCH_PROOF", "STARK_C12_A_PROOF_BATCH_PROOF", "STARK_RECURSIVE_1_PROOF_BATCH_PROOF")

    def run(*extra):
        r = subprocess.run([exe, "-c", cfg, "--proofs", "2"] + list(extra), capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        if r.returncode != 0:
            raise RuntimeError((r.stderr or r.stdout)[-300:])
        return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr
    try:
        pip, log = run()
        seq, _ = run("--sequential-upload")
    except Exception as e:                                   # missing bundles / binary, timeout, ...
        return {"error": repr(e)[:300]}
    t = pip["timers_s"]
    z = {k[len("zkevm."):]: v for k, v in t.items() if k.startswith("zkevm.")}
    grp = lambda key: round(sum(v for k, v in z.items() if key(k)), 4)
    up = seq["timers_s"]["zkevm.STARK_STEP_1_WITNESS_UPLOAD"]
    modes = [m.strip("[] ") for m in log.split("steps: zkevm", 1)[1].splitlines()[0].split("] [")] if "steps: zkevm" in log else None
    out = {
        "metric": "genBatchProof STARK wall-clock (s): the three genProof calls of Prover::genBatchProof (C++: host/zkprover_batch), one GPU, "
                  "zkEVM witness (50.4 GB) uploaded from page-locked host memory inside the wall",
        "wall_s": round(pip["three_starks_s"], 4), "per_stark_s": {k: round(t[k], 4) for k in starks},
        "wall_sequential_upload_s": round(seq["three_starks_s"], 4), "wall_without_upload_s": round(seq["three_starks_s"] - up, 4),
        "witness_upload": {"GB": 50.4, "one_piece_s": round(up, 4), "copy_stream_busy_s_when_pipelined": round(pip["zkevm_witness_upload_ms"] / 1e3, 4),
                           "exposed_s_when_pipelined": round(pip["three_starks_s"] - (seq["three_starks_s"] - up), 4)},
        "same_proof_both_ways": pip["batch_proof_fnv"] == seq["batch_proof_fnv"],
        "zkevm": {"log_n": 23, "wall_s": round(t["STARK_PROOF_BATCH_PROOF"], 4),
                  "stages_s": {"step1_upload_lde_merkle": grp(lambda k: k in ("STARK_STEP_1_LDE_AND_MERKLETREE", "STARK_STEP_1_WITNESS_UPLOAD")), "lde_merkle_2_3": grp(lambda k: k in ("STARK_STEP_2_LDE_AND_MERKLETREE", "STARK_STEP_3_LDE_AND_MERKLETREE")),
                               "lde_kernels_1_2_3": grp(lambda k: k in ("STARK_STEP_1_LDE", "STARK_STEP_2_LDE", "STARK_STEP_3_LDE")),
                               "merkle_kernels_1_2_3": grp(lambda k: k in ("STARK_STEP_1_MERKLETREE", "STARK_STEP_2_MERKLETREE", "STARK_STEP_3_MERKLETREE")),
                               "const_lde": grp(lambda k: "CONST_LDE" in k), "merkle_4": grp(lambda k: k == "STARK_STEP_4_MERKLETREE"),
                               "step_programs": grp(lambda k: "CALCULATE_EXPS" in k and "INTT" not in k),
                               "h1h2_z": grp(lambda k: "H1H2" in k or k.endswith("_Z")), "evmap_lev_xdiv": grp(lambda k: "STEP_5" in k and "EXPS" not in k),
                               "q_intt_ntt": grp(lambda k: "INTT" in k), "fri_and_openings": grp(lambda k: k.endswith("_FRI"))},
                  "timers_s": {k: round(v, 4) for k, v in z.items()}, "programs": modes,
                  "layout": "one-GPU aliased slab (239 GB), device sections column-major (extended ones coset-major)",
                  "batch_proof_fnv": pip["batch_proof_fnv"]},
        "note": "synthetic witnesses and constants (the archive is absent); the executor and the circom witness calculators between the STARKs are "
                "CPU work outside the hot path, replaced by stand-ins outside the timed STARKs",
    }
    # roofline of the step-program leg: algorithmic bytes = every column / table word a program names, once, per row
    try:
        from tools import zkevm_geometry as G
        progs, _ = G.load(23, True, "zkevm")
        names = {"step2prev": "STARK_STEP_2_CALCULATE_EXPS", "step3prev": "STARK_STEP_3_CALCULATE_EXPS", "step3": "STARK_STEP_3_CALCULATE_EXPS_2",
                 "step42ns": "STARK_STEP_4_CALCULATE_EXPS_2NS", "step52ns": "STARK_STEP_5_CALCULATE_EXPS"}
        per, tot_b, tot_t = {}, 0, 0.0
        for st, (code, _, _) in progs.items():
            rd, wr = G.program_io_words(code)
            rows = 1 << (24 if st.endswith("2ns") else 23)
            b = (rd + wr) * 8 * rows
            per[st] = {"words_read_per_row": rd, "words_written_per_row": wr, "alg_GB": round(b / 1e9, 1), "s": round(z[names[st]], 4), "GBs": round(b / z[names[st]] / 1e9, 1)}
            tot_b += b; tot_t += z[names[st]]
        out["roofline_steps"] = {"bound": "hbm", "achieved": round(tot_b / tot_t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(tot_b / tot_t / 1e9 / HBM_PEAK_GBS, 4),
                                 "alg_bytes": tot_b, "seconds": round(tot_t, 4), "per_program": per,
                                 "note": "algorithmic bytes: each column / table word a program reads or writes, once per row (temporaries are not traffic); measured inside the C++ genProof. "
                                         "The metric asks for the HBM fraction; what bounds the largest program (step42ns) is vector issue, not bytes: a quarter fewer bytes through HBM "
                                         "left its time where it was (profiles/r5/step_program_variant_sweep.txt)"}
    except Exception as e:
        out["roofline_steps"] = {"error": repr(e)[:200]}
    return out


def child_leg(tool, tool_args, world, rank, local_rank, port_offset, backend, timeout_s=300):
    """Run tools/<tool> over the ranks of this run as one CHILD process per rank on the rank's GPU, with a rendezvous of its own
    (port_offset above this run's port) and a hard timeout.  The multi-rank exchanges have run with ranks SHARING a GPU only (this
    pool has no multi-GPU box): a failure or a hang on hardware they never saw then costs that leg, not the bench line.
    -> the JSON line rank 0's child printed (rank 0), None (other ranks), or {"error": ...}."""
    import subprocess
    env = dict(os.environ)
    env.update(RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + port_offset), PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    for k in ("TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)                                     # the child makes a plain env:// rendezvous of its own
    cmd = [sys.executable, os.path.join(ROOT, "tools", tool)] + tool_args + ["--backend", backend]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
    try:
        out, err = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        p.kill(); p.communicate()
        return {"error": f"rank {rank}: no result within {timeout_s} s"}
    if p.returncode != 0:
        return {"error": f"rank {rank}: exit {p.returncode}: {err[-300:]}"}
    if rank != 0:
        return None
    try:
        return json.loads(out.strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)[:200] + " / " + out[-200:]}


def genproof_sharded_cpp_leg(world, rank, local_rank, backend, log_n=23, timeout_s=420):
    """The batch proof with the zkEVM STARK ROW-SHARDED over the ranks, through the C++ product path (BASELINE north_star):
    one CHILD process per rank runs host/zkprover_batch --world N --rank r on the rank's GPU -- xlp_comm_init (RCCL over xGMI; "shm" =
    host-staged for ranks sharing a GPU in rehearsals) before any other GPU work, host/starks.hpp genProofSharded on the per-rank window
    bundle (ahead-of-time step programs), c12a / recursive1 on rank 0.  Rank 0's child prints the line; a hard timeout: no multi-GPU node
    has ever been available to this build, so a failure on hardware it never saw costs this leg, not the bench line."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "xlayer_prover_amd", "host", "zkprover_batch")
    bdir = os.path.join(ROOT, "xlayer_prover_amd", "chelpers", "bundles")
    # the windows as COLUMNS (the layout of the one-GPU product path; the only window bundles built since round 5)
    colmajor = True
    bundle = os.path.join(bdir, f"zkevm.geo{log_n}w{world}c")
    built_here = False
    if not os.path.exists(os.path.join(bundle, "zkevm.starkinfo.json")):
        if rank == 0:                                        # (no reference tree needed: the converted programs travel with the repo)
            try:
                from tools import make_stark_bundle as B
                B.make("zkevm", log_n, False, world=world)
                built_here = True
            except Exception as e:
                return {"error": "no window bundle: " + repr(e)[:200]}
    base = json.load(open(os.path.join(ROOT, "tools", "config_runFile_BatchProof_synthetic_colmajor.json")))
    base["zkevmStarkInfo"] = os.path.join(bundle, "zkevm.starkinfo.json")
    port = int(os.environ.get("MASTER_PORT", "29500"))
    cfg = os.path.join(tempfile.gettempdir(), f"xlp_sharded_cfg_{port}_{world}.json")
    tag = os.environ.get("XLP_BENCH_LAUNCH_TAG", str(os.getppid()))     # the same on every rank of this run, different between runs
    rdv = os.path.join(tempfile.gettempdir(), f"xlp_rccl_id_{port}_{world}_{tag}") if backend == "rccl" else f"xlp_bench_{port}_{world}_{tag}"
    if rank == 0:
        json.dump(base, open(cfg + ".tmp", "w")); os.replace(cfg + ".tmp", cfg)
        for stale in ((rdv, rdv + ".tmp") if backend == "rccl" else ("/dev/shm/" + rdv, "/dev/shm/" + rdv + ".ctl")):
            if os.path.exists(stale):                        # (the library does not trust leftovers either: comm.hip)
                os.remove(stale)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()                                       # config (and a bundle built just now) in place before the children start
    cmd = [exe, "-c", cfg, "--proofs", "2", "--timers", "--world", str(world), "--rank", str(rank), "--device", str(local_rank), "--backend", backend, "--rendezvous", rdv]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    try:
        out, err = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        p.kill(); p.communicate()
        return {"error": f"rank {rank}: no result within {timeout_s} s"}
    if p.returncode != 0:
        return {"error": f"rank {rank}: exit {p.returncode}: {err[-300:]}"}
    if rank != 0:
        return None
    try:
        j = json.loads(out.strip().splitlines()[-1])
        t = j["timers_s"]
        z = {k[len("zkevm."):]: round(v, 4) for k, v in t.items() if k.startswith("zkevm.")}
        modes = [m.strip("[] ") for m in err.split("steps: zkevm", 1)[1].splitlines()[0].split("] [")] if "steps: zkevm" in err else None
        return {"metric": f"genBatchProof STARK wall-clock (s), zkEVM STARK row-sharded over {world} rank(s) in the C++ Prover (host/zkprover_batch --world {world}), "
                          "witness rows uploaded per rank inside the wall; c12a / recursive1 on rank 0",
                "wall_s": round(j["three_starks_s"], 4), "zkevm_sharded_s": round(t["STARK_PROOF_BATCH_PROOF"], 4), "c12a_s": round(t["STARK_C12_A_PROOF_BATCH_PROOF"], 4),
                "recursive1_s": round(t["STARK_RECURSIVE_1_PROOF_BATCH_PROOF"], 4), "zkevm_stage_timers_s_rank0": z, "sharded": j.get("sharded"), "programs": modes,
                "window_bundle_built_in_this_run": built_here, "window_layout": "colmajor" if colmajor else "rowmajor", "log_n": log_n, "recursive1_root1": j.get("recursive1_root1"),
                "note": "no run on more than one GPU exists yet: correctness of the multi-rank path is covered by tests/test_cpp_sharded.py (ranks sharing one GPU)"}
    except Exception as e:
        return {"error": repr(e)[:200] + " / " + out[-200:]}


def e2e_leg(timeout_s=300):
    """BASELINE configs[4] through the C++ Prover (host/zkprover_batch --e2e; the SNARK wrap on a synthetic final circuit): two consecutive batch proofs
    (each: zkEVM 2^23 + c12a + recursive1), Prover::genAggregatedProof (joinzkin + the recursive2 STARK) and Prover::genFinalProof up
    to the recursivef STARK (Poseidon-BN128 trees / transcript).  Synthetic witnesses; the circom witness calculators and the fflonk
    prover around them are CPU / out-of-scope work."""
    import subprocess
    exe = os.path.join(ROOT, "xlayer_prover_amd", "host", "zkprover_batch")
    cfg = os.path.join(ROOT, "tools", "config_runFile_e2e_synthetic_colmajor.json")
    try:
        r = subprocess.run([exe, "-c", cfg, "--e2e"], capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        if r.returncode != 0:
            raise RuntimeError((r.stderr or r.stdout)[-300:])
        lines = [json.loads(l) for l in r.stdout.strip().splitlines()]
    except Exception as e:
        return {"error": repr(e)[:300]}
    t = {l["stage"]: l["timers_s"] for l in lines if "timers_s" in l}
    batch = lambda k: round(sum(t[k][n] for n in ("STARK_PROOF_BATCH_PROOF", "STARK_C12_A_PROOF_BATCH_PROOF", "STARK_RECURSIVE_1_PROOF_BATCH_PROOF")), 4)
    agg, fin = t["genAggregatedProof"], t["genFinalProof(recursivef)"]
    out = {"metric": "STARK seconds of batch + batch + aggregated + final(recursivef) in the C++ Prover, one GPU",
           "batch_proof_1_s": batch("genBatchProof#1"), "batch_proof_2_s": batch("genBatchProof#2"),
           "aggregated_recursive2_s": round(agg["STARK_RECURSIVE_2_PROOF_BATCH_PROOF"], 4), "final_recursivef_s": round(fin["STARK_RECURSIVE_F_PROOF_BATCH_PROOF"], 4),
           "recursivef_stages_s": {k[len("recursivef."):]: round(v, 4) for k, v in fin.items() if k.startswith("recursivef.")}}
    out["starks_total_s"] = round(out["batch_proof_1_s"] + out["batch_proof_2_s"] + out["aggregated_recursive2_s"] + out["final_recursivef_s"], 4)
    if "RAPID_SNARK" in fin:                            # the fflonk proof of the final circuit (synthetic key: tools/config_runFile_e2e_synthetic*.json "finalStarkZkey")
        out["final_fflonk_s"] = round(fin["RAPID_SNARK"], 4)
        out["final_fflonk_rounds_s"] = {k[len("fflonk."):]: round(v, 4) for k, v in fin.items() if k.startswith("fflonk.")}
        out["final_fflonk_key"] = "synthetic key of 2^22 rows built in memory (host/fflonk_bench_key.hpp): the cost of a proof, verifiable against nothing; the reference's final.fflonk.zkey is not in the snapshot"
        out["chain_total_s"] = round(out["starks_total_s"] + out["final_fflonk_s"], 4)
    return out


def self_launch(gpus, budget_s):
    """`python3 bench.py --gpus N` started PLAINLY (no torch.distributed.run around it): before anything touches the GPU, start one child per
    rank -- this same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment -- relay rank 0's single JSON line, and leave
    with rank 0's exit code.  (Children, never an exec: a process that has initialised the GPU must not be replaced.)"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    t0 = time.time()
    procs = []
    for r in range(gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   XLP_BENCH_T0=repr(t0), XLP_BENCH_LAUNCH_TAG=f"{os.getpid()}_{int(t0)}")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    line, rc = "", 1
    try:
        out, _ = procs[0].communicate(timeout=budget_s + 150)          # the ranks keep to budget_s themselves; this is the backstop
        rc = procs[0].returncode
        line = out.strip().splitlines()[-1] if out.strip() else ""
    except subprocess.TimeoutExpired:
        line = ""
    for p_ in procs:                                                     # the exact processes started above
        try:
            p_.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p_.kill()
    if not line:
        line = json.dumps({"metric": "Goldilocks NTT GFE/s (LDE 2^%d x %d, blow-up 2)" % (LOG_N, NCOLS), "value": None, "unit": "GFE/s", "n_gpus": gpus,
                           "error": f"rank 0 printed no line (exit {rc}) within {budget_s + 150} s of the self-launched run"})
        rc = rc or 1
    print(line, flush=True)
    return rc


class Budget:
    """ONE deadline for the whole run (the driver gives a bench command 600 s): every leg gets min(its cap, what is left minus what the legs after
    it and the final line need) and is skipped -- named as skipped in the line -- when that is not enough to be worth starting."""

    def __init__(self, seconds):
        self.t0 = float(os.environ.get("XLP_BENCH_T0", time.time()))
        self.seconds = seconds

    def left(self):
        return self.t0 + self.seconds - time.time()

    def grant(self, cap, need, reserve=10.0, dist=None, dev=None, torch=None):
        """seconds this leg may take (0 = skip).  With dist: the SAME answer on every rank (the smallest `left`)."""
        left = self.left()
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.tensor([left], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            left = float(t.item())
        g = min(cap, left - reserve)
        return g if g >= need else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bn128", action="store_true", help="skip the BN128 / fflonk kernel leg (N=1 only; a few seconds)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the batch + batch + aggregated + final(recursivef) chain (N=1 only; ~20 s)")
    ap.add_argument("--no-genproof", action="store_true", help="skip the whole-genProof leg (N=1 only; ~25 s)")
    ap.add_argument("--sharded-commit", type=int, default=-1, help="time the section commit sharded over the ranks (default: on for --gpus > 1; 1 forces it at N=1 "
                    "through RCCL in a world of one)")
    ap.add_argument("--sharded-genproof", type=int, default=-1, help="log2 N of the row-sharded zkEVM-geometry genProof run over the ranks "
                    "(tools/zkevm_genproof_sharded.py, one child process per rank); 0 = off; default: 20 when --gpus > 1, off on one GPU")
    ap.add_argument("--sharded-batchproof", type=int, default=-1, help="log2 N of the batch proof with the zkEVM STARK row-sharded over the ranks through the C++ "
                    "Prover (host/zkprover_batch --world N); 0 = off; default: 23 when --gpus > 1, off on one GPU (opt-in there: a world of one through RCCL)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the --gpus N path with the ranks SHARING GPU 0 "
                    "(exchanges staged through the host); its numbers mean nothing")
    ap.add_argument("--budget", type=float, default=420.0, help="seconds the whole run may take (one global deadline; legs that no longer fit are skipped and "
                    "named in the line)")
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--ncols", type=int, default=NCOLS)
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        sys.exit(self_launch(args.gpus, args.budget))                    # plain `python3 bench.py --gpus N`: be the launcher (no GPU call before this)
    budget = Budget(args.budget)
    # stdout carries exactly ONE line, the JSON: everything else that writes to fd 1 (RCCL prints a version banner there when its
    # first communicator is created) goes to stderr
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from xlayer_prover_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        if args.backend == "gloo":
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=900))
        else:
            dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=900))

    capi.init(local_rank)

    n, ncols = 1 << args.log_n, args.ncols
    n_ext = n << EXT_BITS
    src = torch.empty(n * ncols, dtype=torch.int64, device=dev)
    out = torch.empty(n_ext * ncols, dtype=torch.int64, device=dev)
    # rank r owns columns [8r, 8r+8) of the synthetic trace: element index = row*(8*world) + col
    capi.dev_fill(src, SEED, rank * n * ncols)

    def step():
        capi.dev_extend(out, src, n_ext, n, ncols)

    for _ in range(args.warmup):
        step()
    # per-kernel HIP events (recorded inside the library on the stream the kernels run on)
    torch.cuda.synchronize()
    capi.profile_enable(True)
    for _ in range(10):
        step()
    kern_ms_sum, kern_cnt = capi.profile_read()
    capi.profile_enable(False)
    # the ALU ceiling of those three kernels, measured: their RESIDENT forms (same butterfly networks, twiddle tables, LDS tiles, barriers and grids; the
    # global data traffic removed: xlp_dev_lde_resident, csrc/ntt_lde_common.hpp) on the same geometry, timed the same way
    res_ms_sum, res_cnt = None, 0
    if rank == 0 and args.log_n in (22, 23) and 4 <= ncols <= 64:
        try:
            sink = torch.empty(4096, dtype=torch.int64, device=dev)
            run_res = lambda: capi._ck(capi.load().xlp_dev_lde_resident(capi._dp(sink), n_ext, n, ncols), "xlp_dev_lde_resident")
            run_res()
            torch.cuda.synchronize()
            capi.profile_enable(True)
            for _ in range(10):
                run_res()
            res_ms_sum, res_cnt = capi.profile_read()
            capi.profile_enable(False)
        except Exception as e:                                   # reported in the line, never fatal for the headline
            res_ms_sum, res_cnt = repr(e)[:200], 0

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
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- the exchange-carrying legs.  On one GPU: in this process (opt-in).  On several: one child process per rank, in the order
    # headline (above) -> sharded_commit -> genproof_sharded_cpp, each inside what is left of the ONE budget (a slow first contact with
    # hardware this code has never seen costs the later legs, not the line).  The interpreted Python form of the sharded genProof
    # (tools/zkevm_genproof_sharded.py) is opt-in (--sharded-genproof L): the C++ product path supersedes it.
    sharded, sharded_gp, sharded_bp = None, None, None
    skipped = {}
    gp_log_n = args.sharded_genproof if args.sharded_genproof > 0 else 0
    grant = lambda cap, need, reserve: budget.grant(cap, need, reserve, dist, dev, torch)
    if world == 1 and args.sharded_commit == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29641")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
        del src, out
        torch.cuda.empty_cache()
        try:
            sharded = sharded_commit_leg(capi, torch, dist, dev, world, rank, log_n=args.log_n, ncols=751 if args.log_n == LOG_N else 64)
        except Exception as e:                               # reported, never fatal for the headline line
            sharded = {"error": repr(e)[:300]}
        src = out = None
    # gloo = a rehearsal with the ranks SHARING one GPU: the exchange-carrying legs run at 2^20 rows (two ranks' 2^23-row windows do not fit one GPU)
    rehearsal = args.backend == "gloo" and world > 1
    bp_log_n = args.sharded_batchproof if args.sharded_batchproof >= 0 else ((20 if rehearsal else 23) if world > 1 else 0)
    sc_log_n = 20 if rehearsal and args.log_n == LOG_N else args.log_n
    if world > 1 and (args.sharded_commit != 0 or gp_log_n or bp_log_n):
        src = out = None
        torch.cuda.empty_cache()
        capi.shutdown()                                      # the children get the whole GPU
        if args.sharded_commit != 0:
            g = grant(150.0, 45.0, 150.0 if bp_log_n else 10.0)          # (leave the batch proof, the leg that matters more, its share)
            if g:
                sharded = child_leg("bench_sharded_commit.py", ["--log-n", str(sc_log_n), "--ncols", "751" if sc_log_n == LOG_N else "64", "--steps", "2",
                                                                "--check"], world, rank, local_rank, 29, args.backend, timeout_s=g)
            else:
                skipped["sharded_commit"] = f"{budget.left():.0f} s left of the {args.budget:.0f} s budget"
            dist.barrier()
        if bp_log_n:
            g = grant(330.0, 90.0, 10.0)
            if g:
                sharded_bp = genproof_sharded_cpp_leg(world, rank, local_rank, "rccl" if args.backend == "nccl" else "shm", bp_log_n, timeout_s=g)
            else:
                skipped["genproof_sharded_cpp"] = f"{budget.left():.0f} s left of the {args.budget:.0f} s budget"
            dist.barrier()
        if gp_log_n:
            g = grant(200.0, 60.0, 10.0)
            if g:
                sharded_gp = child_leg("zkevm_genproof_sharded.py", ["--log-n", str(gp_log_n), "--check-one-gpu"], world, rank, local_rank, 23, args.backend, timeout_s=g)
            else:
                skipped["sharded_genproof"] = f"{budget.left():.0f} s left of the {args.budget:.0f} s budget"
            dist.barrier()
    elif gp_log_n:
        src = out = None
        torch.cuda.empty_cache()
        sharded_gp = child_leg("zkevm_genproof_sharded.py", ["--log-n", str(gp_log_n), "--collectives", "--check-one-gpu"], 1, 0, local_rank, 23, "nccl")
    if world == 1 and args.sharded_batchproof > 0:           # opt-in on one GPU: a world of one through RCCL (2^23 does not fit one GPU unaliased: pick <= 22)
        src = out = None
        torch.cuda.empty_cache()
        capi.shutdown()
        sharded_bp = genproof_sharded_cpp_leg(1, 0, local_rank, "rccl", args.sharded_batchproof)
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
        # counter-derived figures: HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE) and VALU instructions of one LDE, from the newest
        # profiles/rN/counters.json; stale = measured on other kernel sources than the ones in the tree
        traffic, valu_issue, traffic_stale, counters_file = None, None, None, None
        cj, stale, counters_file = counters()
        if cj and args.log_n == LOG_N and ncols == NCOLS:
            traffic = cj["lde"]["hbm_bytes_per_lde"]
            traffic_stale = stale
            rate = cj["lde"]["valu_wave_instructions_per_lde"] / (lde_kernel_ms * 1e-3)
            valu_issue = {"achieved": round(rate / 1e9, 1), "unit": "G wave-instr/s", "lane_instructions_per_input_element": round(cj["lde"]["valu_wave_instructions_per_lde"] * 64 / (n * ncols), 1),
                          "note": "instructions retired per second (counter-derived); the ceiling of these kernels is alu_ceiling, measured, not a model of issue rates"}
        alu_ceiling = None
        if res_cnt and kern_cnt:
            res_ms = [x / res_cnt for x in res_ms_sum]
            alu_ceiling = {"kernel": "the resident forms of the same three kernels (xlp_dev_lde_resident): butterfly networks, twiddle tables, LDS tiles, barriers and grids as "
                                     "in the product, global data traffic removed", "avg_launch_ms": {k: round(v, 4) for k, v in zip(LDE_KERNELS, res_ms)},
                           "ms_per_lde": round(sum(res_ms), 4), "GFE_s": round(n_ext * ncols / sum(res_ms) / 1e6, 2),
                           "frac": round(sum(res_ms) / lde_kernel_ms, 4), "per_kernel_frac": {k: round(r / v, 3) for k, r, v in zip(LDE_KERNELS, res_ms, per_kernel_ms)},
                           "note": "frac = resident time / product time: the share of an extendPol that arithmetic and LDS exchanges take by themselves; the rest is what its global loads / stores cost on top"}
        elif isinstance(res_ms_sum, str):
            alu_ceiling = {"error": res_ms_sum}
        roofline = {
            "bound": "valu", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "hbm_frac": round(achieved / HBM_PEAK_GBS, 4), "alu_ceiling": alu_ceiling,
            "traffic": traffic, "traffic_stale": traffic_stale, "counters": counters_file,
            "achieved_from": "sum of the three kernels' average launch durations over 10 event-instrumented steps (the timed region's ms_per_step is the unbracketed figure)",
            "kernel": "xlp::lde_mid_mfma + lde_top_dit (the two largest) + lde_top_dif = one extendPol",
            "launches_per_step": 3,
            "avg_launch_ms": {k: round(v, 4) for k, v in zip(LDE_KERNELS, per_kernel_ms)} if kern_cnt else None,
            "alg_bytes_per_step": alg_bytes_lde,
            "note": "bound by integer vector issue, not by HBM (Goldilocks mul = 4 v_mad_u64_u32 + carry chains on a 32-bit VALU): achieved / peak / frac are the HBM figures "
                    "the metric asks for (algorithmic bytes / kernel time / 8 TB/s), alu_ceiling.frac is how close the kernels are to what bounds them.  Since round 5 the middle pass "
                    "does its 16-point DFTs on the matrix pipe (v_mfma_i32_32x32x32_i8 over byte limbs, in-tile twiddles folded into per-wave matrices: csrc/ntt_lde_mid_mfma.hip); what "
                    "is left on the vector pipe there is the recombination of digit sums and 3 of the former 8 twiddle multiplications per element; the two top passes stay vector kernels "
                    "(their matrix form measured no faster: 64-byte HBM segments bound them)",
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
        if sharded is not None:
            line["sharded_commit"] = sharded
        if sharded_gp is not None:
            line["sharded_genproof"] = sharded_gp
        if sharded_bp is not None:
            line["genproof_sharded_cpp"] = sharded_bp
        if world > 1:
            if rehearsal:
                line["rehearsal"] = "ranks share ONE GPU, exchanges staged through the host (gloo / shm): plumbing evidence, the rates mean nothing"
            line["multi_gpu_evidence"] = ("no run of this code on more than one GPU existed when it was written (none in the build pool): ranks_seen inside "
                                          "genproof_sharded_cpp.sharded is what the transport itself counted in THIS run")
        if world == 1:
            if args.log_n == LOG_N:
                del src, out
                src = out = None
                torch.cuda.empty_cache()
                try:
                    line["roofline_lde_colmajor"] = lde_colmajor_leg(capi, torch, dev, args.log_n)
                except Exception as e:
                    line["roofline_lde_colmajor"] = {"error": repr(e)[:300]}
            line["roofline_merkle"] = merkle_leg(capi, torch, dev)
            if not args.no_bn128:
                line["recursivef_fflonk_kernels"] = bn128_leg(capi, torch, dev)
        if world == 1 and not args.no_genproof and args.log_n == LOG_N:
            del src, out
            torch.cuda.empty_cache()
            capi.shutdown()                                   # the child needs ~300 of the 309 GB: give the library's scratch back too
            g = budget.grant(300.0, 120.0, 60.0)              # (two runs of the binary; the CPU baseline after it needs ~45 s)
            if g:
                line["genproof"] = genproof_leg(timeout_s=g / 2)
            else:
                skipped["genproof"] = f"{budget.left():.0f} s left of the {args.budget:.0f} s budget"
            if not args.no_e2e:
                g = budget.grant(200.0, 60.0, 60.0)
                if g:
                    line["e2e"] = e2e_leg(timeout_s=g)
                else:
                    skipped["e2e"] = f"{budget.left():.0f} s left of the {args.budget:.0f} s budget"
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.log_n, ncols, budget_s=max(3.0, min(20.0, budget.left() / 3)))
            g = genproof_toy_gpu(capi)
            line["cpu_baseline"]["legs"]["genproof_toy_gpu"] = g
            tg = line["cpu_baseline"]["legs"].get("genproof_toy", {})
            if "root1" in g and "root1" in tg:
                line["cpu_baseline"]["legs"]["genproof_toy_same_root"] = g["root1"] == tg["root1"]
        else:
            line["cpu_baseline"] = None
        if skipped:
            line["skipped"] = skipped
        line["seconds_used"] = round(time.time() - budget.t0, 1)
        json_out.write(json.dumps(line) + "\n"); json_out.flush()

    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
