"""The toy prover of tests/mini_genproof_flow.py run ROW-SHARDED over G ranks (one process per GPU), assembled from the
building blocks of xlayer_prover_amd/shard.py -- the multi-GPU form of Starks::genProof (src/starkpil/starks.cpp:9-404):

  stage                       single GPU                      G ranks
  section commit (1,2,3)      extend + merkelize              rows -> columns all-to-all, per-column LDE, columns -> rows
                                                              all-to-all, sub-tree per rank, 32-byte cap all-gather
  step programs               expression VM over all rows     VM over the rank's rows on WINDOWS (rows + halo of the next rank)
  plookup h1 / h2             calculateH1H2                   sharded_h1h2 (all-gather f, t; replicated merge; own rows kept)
  grand product z             calculateZ                      local scan + one Fp^3 per rank (sharded_calculate_z)
  quotient (step 4)           INTT / split / NTT of q         all-gather of the 3 q columns, replicated transform, own rows kept
  LEv / LpEv                  geometric + INTT (N x 3)        replicated, own rows kept
  evmap                       row sum                         local partial sums + all-gather (sharded_evmap)
  FRI                         f_2ns (NE x 3)                  all-gather of f_2ns, FRI replicated (production: rank 0 only)
  queries                     getGroupProof                   owner rank opens the row inside its sub-tree + cap-tree siblings

Every rank runs the transcript itself (all inputs to it are replicated roots / evaluations), so no challenge is ever
broadcast.  The result must equal the single-process proof bit for bit.  Tests only (drives either backend)."""
import numpy as np
import torch
import torch.distributed as dist

from tests import mini_genproof_flow as flow
from xlayer_prover_amd import expr as E
from xlayer_prover_amd import shard

P = flow.P
HALO_N, HALO_E = 1, 2                       # next-row distance on the base / extended domain (1 << extendBits)


class WindowLayout:
    """per-rank slab: every section holds the rank's rows followed by the halo rows of the next rank"""

    def __init__(self, nbits, world):
        self.nbits = nbits
        self.N, self.NE = 1 << nbits, 1 << (nbits + 1)          # GLOBAL sizes: the modulus of the next-row operands
        self.n, self.ne = self.N // world, self.NE // world
        o = 0
        self.off = {}
        for name, w in (("cm1_n", flow.W1), ("cm2_n", flow.W2), ("cm3_n", flow.W3), ("tmpExp_n", flow.WT)):
            self.off[name] = (o, w, self.n + HALO_N); o += w * (self.n + HALO_N)
        for name, w in (("cm1_2ns", flow.W1), ("cm2_2ns", flow.W2), ("cm3_2ns", flow.W3), ("cm4_2ns", flow.W4)):
            self.off[name] = (o, w, self.ne + HALO_E); o += w * (self.ne + HALO_E)
        self.total = o


def prove_sharded(B, nbits, fri_steps, nqueries, seed=1, group=None, collectives=False):
    """B: backend of tests/mini_backends.py (+ to_torch / from_torch).  Returns the same dict as flow.prove on every rank.
    collectives=True takes the exchange paths even in a world of one rank (GPU test of the RCCL plumbing on one GPU)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    shard.FORCE_EXCHANGE = bool(collectives)
    WL = WindowLayout(nbits, world)
    N, NE, n, ne = WL.N, WL.NE, WL.n, WL.ne
    assert ne % HALO_E == 0 and n >= HALO_N and (world & (world - 1)) == 0
    r0, e0 = rank * n, rank * ne
    progs, evmap = flow.programs(WL)
    T, F = B.to_torch, B.from_torch
    w_n, w_ne, shift = B.w(nbits), B.w(nbits + 1), 7
    W1, W2, W3, W4, WC, WT = flow.W1, flow.W2, flow.W3, flow.W4, flow.WC, flow.WT

    # ---- test harness: every rank derives the witness and keeps ITS rows (production: the executor's rows arrive sharded)
    cm1_h, const_h, publics = flow.witness(flow.Layout(nbits), seed)
    pols = B.zeros(WL.total)

    def win(name):
        o, w, rows = WL.off[name]
        return pols[o:o + w * rows]

    def local(name):
        o, w, rows = WL.off[name]
        rl = n if name.endswith("_n") else ne
        return pols[o:o + w * rl]

    def fill_halo(name):
        """the window's halo rows <- first rows of rank + 1 (after the local rows changed)"""
        o, w, rows = WL.off[name]
        rl = n if name.endswith("_n") else ne
        if world == 1:
            B.copy_into(pols[o + w * rl:o + w * rows], pols[o:o + w * (rows - rl)])
            return
        out = shard.halo_exchange(T(local(name)), w, rows - rl, group)
        B.copy_into(pols[o + w * rl:o + w * rows], F(out[w * rl:]))

    B.copy_into(local("cm1_n"), B.from_host(cm1_h.reshape(N, W1)[r0:r0 + n].reshape(-1)))
    fill_halo("cm1_n")
    # ---- setup, replicated (constant polynomials are small next to the trace; their tree is built once per circuit)
    const_full = B.from_host(const_h)
    const_2ns_full = B.zeros(NE * WC)
    B.extend(const_2ns_full, const_full, NE, N, WC)
    nodesC = B.merkle(const_2ns_full, WC, NE)
    take = lambda a, rows, w, first, cnt: np.take(np.asarray(a).reshape(rows, w), np.arange(first, first + cnt) % rows, axis=0).reshape(-1)
    const_n = B.from_host(take(const_h, N, WC, r0, n + HALO_N))
    const_2ns = B.from_host(take(B.to_host(const_2ns_full), NE, WC, e0, ne + HALO_E))
    x_n = B.geometric(n, 1, [pow(w_n, r0, P)], [w_n])
    x_2ns = B.geometric(ne, 1, [shift * pow(w_ne, e0, P) % P], [w_ne])
    sn = pow(shift, N, P)
    zi = B.from_host(np.array([flow._inv((sn * pow(B.w(1), i, P) - 1) % P) for i in range(2)], np.uint64))    # period 2 | ne
    q_2ns = B.zeros(ne * 3); f_2ns = B.zeros(ne * 3)
    challenges = np.zeros(8 * 3, np.uint64)
    evals = np.zeros(len(evmap) * 3, np.uint64)
    xdx = B.zeros(ne * 3); xdwx = B.zeros(ne * 3)

    def run(step, nrows):
        bufs = {"pols": pols, "const_pols": const_n, "const_pols_2ns": const_2ns, "challenges": B.from_host(challenges),
                "public_inputs": B.from_host(publics), "evals": B.from_host(evals), "x_n": x_n, "x_2ns": x_2ns, "zi": zi,
                "x_div_x_sub_xi": xdx, "x_div_x_sub_wxi": xdwx, "q_2ns": q_2ns, "f_2ns": f_2ns}
        B.expr_run(progs[step], bufs, WC, WC, 0, nrows)

    # ---- kernels behind the shard.py call-backs
    def lde(x):                                                    # x: torch [N, wc] -> [NE, wc]
        wc = x.shape[1]
        if wc == 0:
            return x.new_empty((NE, 0))
        dst = B.zeros(NE * wc)
        B.extend(dst, F(x.contiguous().view(-1)), NE, N, wc)
        return T(dst).view(NE, wc)

    def merkle_t(rows):
        return T(B.merkle(F(rows.contiguous().view(-1)), rows.shape[1], rows.shape[0]))

    def hash_pairs(nodes):                                         # the level-1 nodes of a tree over width-4 leaves
        k = nodes.shape[0] // 2
        t = T(B.merkle(F(nodes.contiguous().view(-1)), 4, 2 * k))
        return t[2 * k * 4: 2 * k * 4 + k * 4].view(k, 4).clone()

    trees = {}

    def commit(sec_n, sec_2ns, w):
        """starks.cpp:49-60: extendPol + merkelize of one section, across ranks"""
        if world == 1 and not collectives:
            B.extend(local(sec_2ns), local(sec_n), NE, N, w)
            nodes = T(B.merkle(local(sec_2ns), w, NE))
        else:
            cols = shard.rows_to_columns(T(local(sec_n)).view(n, w), N, group)
            _, nodes, ext_rows = shard.sharded_commit(cols, N, NE, w, group, lde=lde, merkle=merkle_t, hash_pairs=hash_pairs)
            B.copy_into(local(sec_2ns), F(ext_rows.reshape(-1)))
        return finish_tree(sec_2ns, nodes, w)

    def finish_tree(sec_2ns, nodes, w):
        fill_halo(sec_2ns)
        levels = shard.cap_tree(nodes[-4:].clone(), hash_pairs, group)
        trees[sec_2ns] = (nodes, w, levels)
        return [int(v) & 0xFFFFFFFFFFFFFFFF for v in levels[-1][0].tolist()]

    def all_rows(t_local):
        if world == 1 and not collectives:
            return t_local
        return torch.cat(shard._all_gather(t_local, group))

    out = {"publics": [int(v) for v in publics]}
    tr = flow.HostTranscript(B.perm)
    verkey = B.root(nodesC, NE)
    tr.put(verkey); tr.put(publics)
    # ---- 1
    out["root1"] = commit("cm1_n", "cm1_2ns", W1); tr.put(out["root1"])
    # ---- 2
    challenges[0:3] = tr.get_field(); challenges[3:6] = tr.get_field()
    run("step2prev", n)
    fill_halo("tmpExp_n")                                           # step3prev reads t' = t[i + 1]
    f = B.get_cols(local("tmpExp_n"), n, WT, 0, 3); t = B.get_cols(local("tmpExp_n"), n, WT, 3, 3)

    def h1h2_full(ff, tt):
        rows = ff.numel() // 3
        h1 = B.zeros(rows * 3); h2 = B.zeros(rows * 3)
        B.h1h2(h1, h2, F(ff), F(tt), rows, 3)
        return T(h1), T(h2)
    h1, h2 = shard.sharded_h1h2(T(f), T(t), 3, h1h2_full, group)
    B.set_cols(local("cm2_n"), n, W2, 0, 3, F(h1)); B.set_cols(local("cm2_n"), n, W2, 3, 3, F(h2))
    fill_halo("cm2_n")                                              # step3prev reads h1' = h1[i + 1]
    out["root2"] = commit("cm2_n", "cm2_2ns", W2); tr.put(out["root2"])
    # ---- 3
    challenges[6:9] = tr.get_field(); challenges[9:12] = tr.get_field()
    run("step3prev", n)
    num = B.get_cols(local("tmpExp_n"), n, WT, 6, 3); den = B.get_cols(local("tmpExp_n"), n, WT, 9, 3)

    def calc_z(a, b):
        z = B.zeros(n * 3)
        B.calc_z(z, F(a), F(b), n)
        return T(z)
    scale = E.Program()
    scale.mul(E.pols(0, 3, 3), E.pols(0, 3, 3), E.challenge(0))

    def scale3(z, s):                                               # z *= s, one expression-program launch
        B.expr_run(scale, {"pols": F(z), "challenges": B.from_host(np.array(s, np.uint64))}, 0, 0, 0, n)
    z = shard.sharded_calculate_z(T(num), T(den), calc_z, scale3, group)
    B.set_cols(local("cm3_n"), n, W3, 0, 3, F(z))
    fill_halo("cm3_n")
    run("step3", n)
    out["root3"] = commit("cm3_n", "cm3_2ns", W3); tr.put(out["root3"])
    # ---- 4: the quotient.  q_2ns is 3 columns: gather them, transform replicated, keep my rows of cm4_2ns
    challenges[12:15] = tr.get_field()
    run("step42ns", ne)
    q_full = F(all_rows(T(q_2ns)))
    qq1 = B.zeros(NE * flow.QDIM)
    B.ntt(qq1, q_full, NE, flow.QDIM, True)
    qq2 = B.q_split(qq1, N, NE, flow.QDEG, flow.QDIM, pow(flow._inv(shift), N, P))
    cm4_full = B.zeros(NE * W4)
    B.ntt(cm4_full, qq2, NE, flow.QDIM * flow.QDEG, False)
    B.copy_into(local("cm4_2ns"), cm4_full[e0 * W4:(e0 + ne) * W4])
    nodes4 = T(B.merkle(local("cm4_2ns"), W4, ne))
    out["root4"] = finish_tree("cm4_2ns", nodes4, W4); tr.put(out["root4"])
    # ---- 5
    xi = tr.get_field(); challenges[21:24] = xi
    xis = flow._mul3s(xi, flow._inv(shift)); wxis = flow._mul3s(flow._mul3s(xi, w_n), flow._inv(shift))
    LEv = B.geometric(N, 3, [1, 0, 0], xis); LpEv = B.geometric(N, 3, [1, 0, 0], wxis)
    B.ntt(LEv, LEv, N, 3, True); B.ntt(LpEv, LpEv, N, 3, True)
    LEv_l = B.from_host(B.to_host(LEv)[3 * r0:3 * (r0 + n)]); LpEv_l = B.from_host(B.to_host(LpEv)[3 * r0:3 * (r0 + n)])
    ents = []
    for s, c, d, pr in evmap:
        ents.append((const_2ns, WC, c, d, pr) if s == "const" else (win(s), WL.off[s][1], c, d, pr))
    part = np.asarray(B.evmap(LEv_l, LpEv_l, n, 1, ents), np.uint64).reshape(-1)
    ev = shard.sharded_evmap(torch.from_numpy(part.view(np.int64).copy()), group)
    evals[:] = np.array(ev, np.uint64).reshape(-1)
    out["evals"] = [[int(v) for v in row] for row in ev]
    for row in out["evals"]:
        tr.put(row)
    challenges[15:18] = tr.get_field(); challenges[18:21] = tr.get_field()
    B.copy_into(xdx, B.x_div(x_2ns, xi, ne)); B.copy_into(xdwx, B.x_div(x_2ns, flow._mul3s(xi, w_n), ne))
    run("step52ns", ne)
    # ---- FRI on the gathered polynomial; step-0 openings come from the owning ranks
    f_full = F(all_rows(T(f_2ns)))

    def open_step0(ys):
        mine = {}
        for qi, y in enumerate(ys):
            for si, name in enumerate(("cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns")):
                nodes, w, levels = trees[name]
                pr = shard.sharded_merkle_proof(y, ne, lambda li: B.proof(local(name), F(nodes), w, ne, li), levels, rank)
                if pr is not None:
                    mine[(qi, si)] = pr
        if world > 1 or collectives:
            parts = [None] * world
            dist.all_gather_object(parts, mine, group=group)
            mine = {k: v for p_ in parts for k, v in p_.items()}
        return [[mine[(qi, si)] for si in range(4)] + [B.proof(const_2ns_full, nodesC, WC, NE, y)] for qi, y in enumerate(ys)]
    out["fri"] = flow.fri_prove(B, tr, f_full, fri_steps, nqueries, shift, open_step0)
    out["verkey"] = verkey
    return out
