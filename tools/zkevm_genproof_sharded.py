#!/usr/bin/env python3
"""Starks::genProof (src/starkpil/starks.cpp:9-404) at the fork-9 zkEVM geometry ROW-SHARDED over G ranks, one process per GPU
(torch.distributed: RCCL over xGMI; gloo with ranks sharing a GPU in the tests) -- the multi-GPU form of
tools/bench_zkevm_genproof.py, assembled from xlayer_prover_amd/shard.py:

  stage                        one GPU                           G ranks
  section commit (1, 2, 3)     extendPol + merkelize             rows -> columns all-to-all, per-column LDE, columns -> rows all-to-all,
                                                                 sub-tree per rank, 32-byte cap all-gather
  step programs                VM over all rows                  VM over the rank's WINDOW (its rows + the halo rows of the next rank; the
                                                                 programs are relocated to the window slab and keep the GLOBAL modulus in
                                                                 their next-row operands); next-row STORES (step3prev, step3) go back to
                                                                 their owner (halo_writeback)
  plookup h1 / h2              calculateH1H2                     t all-gathered, f / h1 / h2 replicated, own rows kept
  grand products z             calculateZ                        local scan + one Fp^3 per rank (sharded_calculate_z)
  quotient (step 4)            INTT / split / NTT of q           q (3 columns) all-gathered, transforms replicated, own rows of cm4 kept
  LEv / LpEv                   geometric + INTT (N x 3)          replicated, own rows used
  evmap                        row sums                          local partial sums + all-gather (sharded_evmap)
  FRI                          f_2ns (NE x 3)                    f all-gathered, FRI replicated; step-0 openings by the owning rank
Every rank runs the transcript itself (its inputs are replicated roots / evaluations), so no challenge is ever broadcast.
With the same synthetic witness, constants and plookup selections the roots, evaluations and FRI layers equal the one-GPU run's.

    python -m torch.distributed.run --nproc-per-node G tools/zkevm_genproof_sharded.py --log-n 18
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from tools import zkevm_geometry as G
from tools.bench_zkevm_genproof import P, evmap_from_step52ns, gl_w, inv
from xlayer_prover_amd import capi, shard
from xlayer_prover_amd import expr as E
from xlayer_prover_amd.transcript import HostTranscript


def _next_row_stores(code, L):
    """{section: [(first column, columns, next distance)]} of the program's stores to next rows"""
    out = {}
    for name, col, dim, nxt in G.next_row_stores(code, L):
        out.setdefault(name, []).append((col, dim, nxt))
    return out


def prove_sharded(ln, group=None, collectives=False, queries=0, fri_steps=None, lookups=-1, zs=-1, device="cuda", timers=None):
    """-> {"roots", "fri_roots_and_head_of_final_pol", "evals", "queries", "openings"} on every rank (the first three are what
    tools/bench_zkevm_genproof.py prove() reports for the same geometry)"""
    circ = "zkevm"
    C = G.CIRCUITS[circ]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    shard.FORCE_EXCHANGE = bool(collectives)
    ext = C["ext"]
    lne = ln + ext
    N, NE = 1 << ln, 1 << lne
    n, ne = N // world, NE // world
    HN, HE = 1, 1 << ext
    assert (world & (world - 1)) == 0 and ne % HE == 0 and n >= HN
    r0, e0 = rank * n, rank * ne
    queries = queries or C["queries"]
    steps = fri_steps or [lne] + [b for b in (lne - 5, lne - 10, lne - 14, lne - 18) if b >= 3]
    secs = G.sections(circ)
    L = G.layout_window(ln, world, circ)
    progs, info = G.load(ln, False, circ, new_layout=L)
    nconst = max(info["n_const"], info["n_const2"])
    W = {name: w for name, w, _ in secs}
    i64 = torch.int64
    dev = device

    def tm(name, fn):
        if timers is None:
            return fn()
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        timers[name] = timers.get(name, 0.0) + time.perf_counter() - t0
        return r
    pols = torch.zeros(L["end"][0], dtype=i64, device=dev)

    def win(name):
        o, w, rows = L[name]
        return pols[o:o + w * rows]

    def local(name):
        o, w, rows = L[name]
        return pols[o:o + w * (ne if name.endswith("_2ns") else n)]

    def fill_halo(name):
        o, w, rows = L[name]
        if w == 0:
            return
        rl = ne if name.endswith("_2ns") else n
        out = shard.halo_exchange(local(name), w, rows - rl, group)
        pols[o + w * rl:o + w * rows] = out[w * rl:]

    def all_rows(t_local):
        if world == 1 and not collectives:
            return t_local
        return torch.cat(shard._all_gather(t_local.contiguous(), group))
    # ---- the witness: every rank derives ITS rows of the synthetic trace (production: the executor's rows arrive sharded)
    capi.dev_fill(local("cm1_n"), 0x5EED0401, r0 * W["cm1_n"])
    fill_halo("cm1_n")
    # ---- setup, replicated: constants, their extension and tree (once per circuit), x tables, ZhInv
    const_full = torch.empty(N * nconst, dtype=i64, device=dev); capi.dev_fill(const_full, 0x5EED0402, 0)
    const_2ns_full = torch.empty(NE * nconst, dtype=i64, device=dev)
    capi.dev_extend(const_2ns_full, const_full, NE, N, nconst)
    nodesC = torch.empty((2 * NE - 1) * 4, dtype=i64, device=dev)
    capi.dev_merkle(nodesC, const_2ns_full, nconst, NE)
    take = lambda full, rows, first, cnt: full.view(rows, nconst)[(torch.arange(first, first + cnt, device=dev) % rows)].contiguous().view(-1)
    const_n = take(const_full, N, r0, n + HN)
    const_2ns = take(const_2ns_full, NE, e0, ne + HE)
    w_n, w_ne, shift = gl_w(ln), gl_w(lne), 7
    x_n = torch.empty(n, dtype=i64, device=dev); capi.dev_geometric(x_n, n, 1, [pow(w_n, r0, P)], [w_n])
    x_2ns = torch.empty(ne, dtype=i64, device=dev); capi.dev_geometric(x_2ns, ne, 1, [shift * pow(w_ne, e0, P) % P], [w_ne])
    sn = pow(shift, N, P)
    zi = capi.from_np(np.array([inv((sn * pow(gl_w(ext), i, P) - 1) % P) for i in range(1 << ext)], np.uint64))
    xdx = torch.empty(ne * 3, dtype=i64, device=dev); xdwx = torch.empty(ne * 3, dtype=i64, device=dev)
    challenges = np.zeros(8 * 3, np.uint64)
    publics = np.arange(1, info["npub"] + 1, dtype=np.uint64)
    evals_d = torch.zeros(3 * info["nev"], dtype=i64, device=dev)
    vm = {s: capi.ExprProgram(code, n1, n3) for s, (code, n1, n3) in progs.items()}
    stores = {s: _next_row_stores(code, L) for s, (code, _, _) in progs.items()}
    ents, ev_ids = evmap_from_step52ns(progs["step52ns"][0], L, const_2ns, nconst, pols, secs)
    ev_arr = capi.evmap_entries(ents)
    ev_out = torch.zeros(3 * len(ents), dtype=i64, device=dev)
    ev_index = torch.tensor([3 * k + j for k in ev_ids for j in range(3)], device=dev)
    lk_picks, z_picks = G.lookup_picks(circ, progs, L)
    if lookups >= 0:
        lk_picks = lk_picks[:lookups]
    if zs >= 0:
        z_picks = z_picks[:zs]
    qdeg = W["cm4_n"] // 3

    def run(step, nrows):
        sp = capi.steps_params(n_const_pols=nconst, n_const_pols_2ns=nconst, pols=pols, const_pols=const_n, const_pols_2ns=const_2ns,
                               challenges=capi.from_np(challenges), public_inputs=capi.from_np(publics), evals=evals_d, x_n=x_n, x_2ns=x_2ns,
                               zi=zi, x_div_x_sub_xi=xdx, x_div_x_sub_wxi=xdwx, q_2ns=local("q_2ns"), f_2ns=local("f_2ns"))
        vm[step].run(sp, 0, nrows)
        if world > 1:                                        # (one rank: the store wraps to row 0 by the operand's own modulus)
            for name, cols in stores[step].items():          # rows written past the window's end belong to the next rank
                o, w, rows = L[name]
                shard.halo_writeback(win(name), w, n, rows - n, cols, group)

    trees = {}

    def finish_tree(sec_2ns, nodes, w):
        fill_halo(sec_2ns)
        levels = shard.cap_tree(nodes[-4:].clone(), shard.gpu_hash_pairs, group)
        trees[sec_2ns] = (nodes, w, levels)
        return [int(v) & 0xFFFFFFFFFFFFFFFF for v in levels[-1][0].tolist()]

    def commit(sec_n, sec_2ns):
        """starks.cpp:49-60: extendPol + merkelize of one section, across ranks"""
        w = W[sec_n]
        if world == 1 and not collectives:
            if w:
                capi.dev_extend(local(sec_2ns), local(sec_n), NE, N, w)
            nodes = torch.empty((2 * NE - 1) * 4, dtype=i64, device=dev)
            capi.dev_merkle(nodes, local(sec_2ns), w, NE)
        else:
            cols = shard.rows_to_columns(local(sec_n).view(n, w), N, group)
            _, nodes, ext_rows = shard.sharded_commit(cols, N, NE, w, group)
            local(sec_2ns).copy_(ext_rows.reshape(-1))
        return finish_tree(sec_2ns, nodes, w)

    out = {}
    tr = HostTranscript(capi.perm)
    verkey = [int(v) for v in capi.to_np(nodesC[-4:])]
    tr.put(verkey); tr.put(publics)
    # ---- 1
    roots = [tm("STARK_STEP_1_LDE_AND_MERKLETREE", lambda: commit("cm1_n", "cm1_2ns"))]; tr.put(roots[-1])
    # ---- 2
    challenges[0:3] = tr.get_field(); challenges[3:6] = tr.get_field()
    tm("STARK_STEP_2_CALCULATE_EXPS", lambda: run("step2prev", n))
    wt, w2, w3 = W["tmpExp_n"], W["cm2_n"], W["cm3_n"]

    def lookups_():
        t_l = torch.empty(3 * n, dtype=i64, device=dev)
        f_pl = torch.empty(3 * N, dtype=i64, device=dev); h1 = torch.empty(3 * N, dtype=i64, device=dev); h2 = torch.empty(3 * N, dtype=i64, device=dev)
        for k, c in enumerate(lk_picks):
            capi.dev_copy_cols(t_l, 3, local("tmpExp_n")[c:], wt, n, 3)                  # transposeH1H2Columns on my rows
            t_pl = all_rows(t_l)                                                            # the table, replicated (2 x N x 3 words per lookup)
            capi.dev_gather_rows(f_pl, t_pl, 3, N, 3, 0x5EED0500 + k)
            capi.dev_calculate_h1h2(h1, h2, f_pl, t_pl, N, 3)
            capi.dev_copy_cols(local("cm2_n")[6 * k:], w2, h1[3 * r0:], 3, n, 3)
            capi.dev_copy_cols(local("cm2_n")[6 * k + 3:], w2, h2[3 * r0:], 3, n, 3)
    tm("STARK_STEP_2_CALCULATEH1H2", lookups_)
    fill_halo("tmpExp_n"); fill_halo("cm2_n"); fill_halo("cm3_n")
    roots.append(tm("STARK_STEP_2_LDE_AND_MERKLETREE", lambda: commit("cm2_n", "cm2_2ns"))); tr.put(roots[-1])
    # ---- 3
    challenges[6:9] = tr.get_field(); challenges[9:12] = tr.get_field()
    tm("STARK_STEP_3_CALCULATE_EXPS", lambda: run("step3prev", n))
    scale = E.Program()
    scale.mul(E.pols(0, 3, 3), E.pols(0, 3, 3), E.challenge(0))
    scale_vm = capi.ExprProgram(scale.code, scale.ntmp1, scale.ntmp3)

    def zs_():
        num = torch.empty(3 * n, dtype=i64, device=dev); den = torch.empty(3 * n, dtype=i64, device=dev)

        def calc_z(a, b):
            z = torch.empty(3 * n, dtype=i64, device=dev)
            capi.dev_calculate_z(z, a, b, n)
            return z

        def scale3(z, s):
            scale_vm.run(capi.steps_params(pols=z, challenges=capi.from_np(np.array(s, np.uint64))), 0, n)
        for k, (cn, cd) in enumerate(z_picks):
            capi.dev_copy_cols(num, 3, local("tmpExp_n")[cn:], wt, n, 3); capi.dev_copy_cols(den, 3, local("tmpExp_n")[cd:], wt, n, 3)
            z = shard.sharded_calculate_z(num, den, calc_z, scale3, group)
            capi.dev_copy_cols(local("cm3_n")[3 * k:], w3, z, 3, n, 3)
    tm("STARK_STEP_3_CALCULATE_Z", zs_)
    fill_halo("tmpExp_n"); fill_halo("cm3_n")
    tm("STARK_STEP_3_CALCULATE_EXPS_2", lambda: run("step3", n))
    roots.append(tm("STARK_STEP_3_LDE_AND_MERKLETREE", lambda: commit("cm3_n", "cm3_2ns"))); tr.put(roots[-1])
    # ---- 4: the quotient.  q_2ns is 3 columns: gather them, transform replicated, keep my rows of cm4_2ns
    challenges[12:15] = tr.get_field()
    tm("STARK_STEP_4_CALCULATE_EXPS_2NS", lambda: run("step42ns", ne))

    def q_poly():
        q_full = all_rows(local("q_2ns"))
        qq1 = torch.empty(3 * NE, dtype=i64, device=dev); qq2 = torch.empty(3 * qdeg * NE, dtype=i64, device=dev)
        cm4_full = torch.empty(3 * qdeg * NE, dtype=i64, device=dev)
        capi.dev_ntt(qq1, q_full, NE, 3, inverse=True)
        capi.dev_q_split(qq2, qq1, N, NE, qdeg, 3, pow(inv(shift), N, P))
        capi.dev_ntt(cm4_full, qq2, NE, 3 * qdeg)
        w4 = 3 * qdeg
        local("cm4_2ns").copy_(cm4_full[e0 * w4:(e0 + ne) * w4])
    tm("STARK_STEP_4_CALCULATE_EXPS_2NS_INTT_MUL_NTT", q_poly)

    def tree4():
        nodes = torch.empty((2 * ne - 1) * 4, dtype=i64, device=dev)
        capi.dev_merkle(nodes, local("cm4_2ns"), 3 * qdeg, ne)
        return finish_tree("cm4_2ns", nodes, 3 * qdeg)
    roots.append(tm("STARK_STEP_4_MERKLETREE", tree4)); tr.put(roots[-1])
    # ---- 5
    xi = tr.get_field(); challenges[21:24] = xi
    mul3s = lambda v, s: [x * s % P for x in v]

    def evmap():
        LEv = torch.empty(3 * N, dtype=i64, device=dev); LpEv = torch.empty(3 * N, dtype=i64, device=dev)
        capi.dev_geometric(LEv, N, 3, [1, 0, 0], mul3s(xi, inv(shift))); capi.dev_geometric(LpEv, N, 3, [1, 0, 0], mul3s(mul3s(xi, w_n), inv(shift)))
        capi.dev_ntt(LEv, LEv, N, 3, inverse=True); capi.dev_ntt(LpEv, LpEv, N, 3, inverse=True)
        capi.dev_evmap(ev_out, LEv[3 * r0:3 * (r0 + n)].contiguous(), LpEv[3 * r0:3 * (r0 + n)].contiguous(), n, ext, ev_arr)
        ev = shard.sharded_evmap(ev_out, group)
        flat = np.array([v for row in ev for v in row], np.uint64)
        evals_d[ev_index] = capi.from_np(flat)
        return [int(v) for v in flat]
    ev = tm("STARK_STEP_5_EVMAP", evmap)
    tr.put(ev)
    challenges[15:18] = tr.get_field(); challenges[18:21] = tr.get_field()
    capi.dev_x_div_x_sub_xi(xdx, x_2ns, xi, ne); capi.dev_x_div_x_sub_xi(xdwx, x_2ns, mul3s(xi, w_n), ne)
    tm("STARK_STEP_5_CALCULATE_EXPS", lambda: run("step52ns", ne))
    # ---- FRI on the gathered polynomial (3 columns); step-0 openings come from the owning ranks
    fri_roots = []

    def fri():
        pol = all_rows(local("f_2ns")); bits = steps[0]; sinv = inv(shift)
        ftrees = []
        for si, nb in enumerate(steps):
            red = bits - nb
            x = tr.get_field()
            pol2 = torch.empty(3 << nb, dtype=i64, device=dev)
            capi.dev_fri_fold(pol2, pol, bits, red, x, sinv)
            if si < len(steps) - 1:
                ng = 1 << steps[si + 1]; gs = (1 << nb) // ng
                aux = torch.empty(3 << nb, dtype=i64, device=dev)
                capi.dev_transpose3(aux, pol2, 1 << nb, steps[si + 1])
                nodes = torch.empty((2 * ng - 1) * 4, dtype=i64, device=dev)
                capi.dev_merkle(nodes, aux, gs * 3, ng)
                ftrees.append((aux, nodes, gs * 3, ng))
                fri_roots.append([int(v) for v in capi.to_np(nodes[-4:])]); tr.put(fri_roots[-1])
            else:
                final_pol = [int(v) for v in capi.to_np(pol2)]
                fri_roots.append(final_pol[:6]); tr.put(final_pol)
            pol = pol2; bits = nb
            for _ in range(red):
                sinv = sinv * sinv % P
        ys = [int(y) for y in tr.get_permutations(queries, steps[0])]
        # step-0 openings: the owner of row y opens it inside its sub-tree and appends the cap-tree siblings
        mine = {}
        for name in ("cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns"):
            nodes, w, levels = trees[name]
            if world == 1 and not collectives:
                rows_here = NE
            else:
                rows_here = ne
            own = [(qi, y) for qi, y in enumerate(ys) if y // rows_here == (rank if rows_here == ne else 0)]
            if not own or w == 0:
                continue
            pr = capi.dev_merkle_proofs(local(name), nodes, w, rows_here, [y % rows_here for _, y in own])
            for (qi, y), p_ in zip(own, pr):
                row = [int(v) for v in p_[:w]]
                sib = [[int(v) for v in p_[w + 4 * l: w + 4 * l + 4]] for l in range((len(p_) - w) // 4)]
                i = y // rows_here
                for lvl in levels[:-1]:
                    sib.append([int(v) & 0xFFFFFFFFFFFFFFFF for v in lvl[i ^ 1].tolist()]); i //= 2
                mine[(qi, name)] = (row, sib)
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, mine, group=group)
            mine = {k: v for p_ in parts for k, v in p_.items()}
        return ys, mine
    ys, openings = tm("STARK_STEP_FRI", fri)
    for o in list(vm.values()) + [scale_vm]:
        o.close()
    return {"roots": roots, "fri_roots_and_head_of_final_pol": fri_roots, "evals": ev, "queries": ys, "openings": openings, "verkey": verkey,
            "widths": {k: W[k] for k in ("cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns")}, "steps": steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--queries", type=int, default=0)
    ap.add_argument("--collectives", action="store_true", help="take the exchange paths even in a world of one rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: the ranks share GPU 0, exchanges staged through the host (rehearsal)")
    ap.add_argument("--check-one-gpu", action="store_true", help="rank 0 then runs the one-GPU flow (tools/bench_zkevm_genproof.py, programs "
                    "interpreted as here) on the same synthetic witness: same roots / FRI layers expected, its wall reported beside the sharded one")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if a.backend == "nccl" else 0
    torch.cuda.set_device(local_rank)
    if a.backend == "gloo":
        dist.init_process_group("gloo")
    elif world > 1 or a.collectives:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    capi.init(local_rank); capi.use_torch_stream()
    res, timers = None, {}
    for rep in range(2):                                   # pass 0 warms allocators / code objects
        timers.clear()
        shard.EXCHANGE_STATS = {} if dist.is_initialized() else None
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        t0 = time.perf_counter()
        res = prove_sharded(a.log_n, collectives=a.collectives, queries=a.queries, timers=timers)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    if not dist.is_initialized() or dist.get_rank() == 0:
        line = {"workload": f"zkEVM-geometry genProof row-sharded over {world} rank(s), N = 2^{a.log_n}; step programs interpreted "
                            "(window-relocated programs have no ahead-of-time plugins yet)",
                "wall_s_incl_setup": round(wall, 4), "sum_timed_s": round(sum(timers.values()), 4),
                "timers_s": {k: round(v, 4) for k, v in timers.items()}, "roots": res["roots"], "fri_roots_and_head_of_final_pol": res["fri_roots_and_head_of_final_pol"]}
        st = shard.EXCHANGE_STATS
        if st and st.get("exchange_s"):
            line["exchange_rank0"] = {"exchange_calls": st["exchange_calls"], "exchange_s": round(st["exchange_s"], 4),
                                      "GB_sent_to_peers": round(st["bytes_sent_to_peers"] / 1e9, 3),
                                      "GBs_sent_per_rank": round(st["bytes_sent_to_peers"] / 1e9 / st["exchange_s"], 1)}
        if a.check_one_gpu:
            import argparse as _ap
            from tools import bench_zkevm_genproof as one
            expect = (res["roots"], res["fri_roots_and_head_of_final_pol"])
            res = None
            torch.cuda.empty_cache()
            o = one.prove(_ap.Namespace(log_n=a.log_n, circuit="zkevm", batch=False, fri_steps="", queries=a.queries, lookups=-1, zs=-1, no_jit=True,
                                        no_dots=False, vm_only="", no_opt=False, aliased=0), "zkevm", a.log_n)
            line["one_gpu_same_programs"] = {"genproof_wall_s": o["genproof_wall_s"], "timers_s": o["timers_s"],
                                             "same_roots_and_fri_layers": (o["roots"], o["fri_roots_and_head_of_final_pol"]) == expect,
                                             "same_by_tree": [x == y for x, y in zip(o["roots"] + o["fri_roots_and_head_of_final_pol"], expect[0] + expect[1])]}
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()                                     # (the other ranks wait for rank 0's one-GPU re-run)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
