"""Oracle checks of a full-size Starks::genProof run (tools/bench_zkevm_genproof.py), driven through the tool's hook interface.
TEST INFRASTRUCTURE: the only place where the big run meets the CPU oracle; the tool itself never imports `oracle`.

What is checked on the run (at N = 2^23 the full buffers cannot be recomputed on a CPU, so everything is sampled):
  * step programs -- for each of the five programs, windows of consecutive rows (incl. the wrap-around N-1 -> 0) are gathered
    from every section before the GPU step, the AS-CONVERTED reference program (tools/chelpers_to_xlp.py output, relocated to
    the window) runs on the oracle, and the rows the GPU left behind -- whatever form it ran: dot-product rewrite, ahead-of-time
    plugin, table program + xlp_dev_dot_columns -- must equal the oracle's rows word for word, in every section;
  * evaluations -- Starks::evmap (starks.cpp:556-669) restated on the oracle (LEv / LpEv by geometric + INTT, then the strided
    dot product) for sampled entries of the evaluation map, from columns downloaded from the extended sections;
  * Merkle openings -- every opened row + sibling path of the five trees verifies against the root the transcript absorbed
    (merkleTreeGL.cpp:12-35 walked by the oracle's permutation), for all queries;
  * FRI -- the first FRI tree opens the FRI polynomial itself (f_2ns at the query positions), and for every query and step the
    oracle's fold of the opened group (friProve.cpp:44-108) equals the value opened in the next tree / the final polynomial;
    the final polynomial's degree is < 2^(last nBits) / blow-up (INTT on the oracle);
  * transcript -- replayed on the oracle from the roots / evaluations / FRI data: same challenges, same query indices."""
import numpy as np

from tools import zkevm_geometry as G
from xlayer_prover_amd import expr as E
from xlayer_prover_amd import programs as R

P = 0xFFFFFFFF00000001


class GenProofChecker:
    def __init__(self, orc, windows=16, K=8, neval=10, seed=7):
        self.orc, self.windows, self.K0, self.K, self.neval = orc, windows, K, K, neval
        self.rng = np.random.default_rng(seed)
        self.report = {"step_program_rows": {}, "evals_checked": 0, "merkle_paths": 0, "fri_folds": 0, "fri_first_tree_vs_f": 0}
        self._pre = None

    # ------------------------------------------------------------------ hooks called by the tool
    def begin(self, env):
        self.env = env

    def pre_step(self, step, nrows):
        """gather the windows of every live section (and the per-row tables) before the GPU evaluates `step`"""
        env = self.env
        base = nrows == env.N
        nxt = 1 if base else (1 << env.ext)
        K = max(self.K0, 2 * nxt + 4)               # rows nxt .. K-nxt-1 of a window are compared
        self.K = K
        starts = [0, nrows - K // 2, nrows // 2 - 1] + [int(v) for v in self.rng.integers(0, nrows - K, self.windows - 3)]
        wins = [[(s + k) % nrows for k in range(K)] for s in starts]
        self._pre = {"step": step, "nrows": nrows, "nxt": nxt, "wins": wins, "snap": [self._gather(w, base) for w in wins]}

    def post_step(self, step, nrows):
        pre = self._pre
        assert pre["step"] == step
        env, orc, K, nxt = self.env, self.orc, self.K, pre["nxt"]
        base = nrows == env.N
        code, n1, n3 = R.read_program(self._program_file(step))
        secs = self._sections(base)
        # window layout: the live sections packed one after the other, K rows each
        mini, o = {}, 0
        for name, w in secs:
            mini[name] = (o, w, K); o += w * K
        mini_code = self._relocate(code, mini)
        checked = 0
        for win, snap in zip(pre["wins"], pre["snap"]):
            bufs = self._oracle_buffers(snap, secs, o, base)
            orc.expr_run(mini_code, n1, n3, bufs, 0, K - nxt, env.nconst, env.nconst)
            after = self._gather(win, base)
            for name, w in secs:
                mo = mini[name][0]
                exp = bufs["pols"][mo:mo + w * K].reshape(K, w)
                got = after[name]
                # rows nxt .. K-nxt-1 were evaluated AND received the next-row stores of an evaluated predecessor
                assert (exp[nxt:K - nxt] == got[nxt:K - nxt]).all(), f"{step}: section {name} differs from the oracle at rows {win[nxt:K - nxt]}"
            checked += K - 2 * nxt
        self.report["step_program_rows"][step] = checked
        self._pre = None

    def evals(self, ents, ev, xi, w_n):
        """ents: the tool's evaluation map [(section tensor, width, column, dim, prime)], ev: the GPU's evaluations (flat)"""
        env, orc = self.env, self.orc
        N, ext = env.N, env.ext
        sinv = orc.inv(7)
        m3 = lambda v, s: [orc.mul(int(x), s) for x in v]
        LEv = orc.ntt(orc.geometric(N, 3, [1, 0, 0], m3(xi, sinv)), N, 3, inverse=True)
        LpEv = orc.ntt(orc.geometric(N, 3, [1, 0, 0], m3(m3(xi, w_n), sinv)), N, 3, inverse=True)
        ev = np.asarray(ev, np.uint64).reshape(-1, 3)
        n = len(ents)
        pick = {0, n - 1}
        # one of every (dim, prime, constant-or-not) combination present, then random ones
        seen = set()
        for i, (t, width, col, dim, prime) in enumerate(ents):
            key = (dim, bool(prime), t.data_ptr() == env.const_2ns.data_ptr())
            if key not in seen:
                seen.add(key); pick.add(i)
        while len(pick) < min(self.neval, n):
            pick.add(int(self.rng.integers(0, n)))
        for i in sorted(pick):
            t, width, col, dim, prime = ents[i]
            colv = env.to_np(t.view(-1, width)[:, col:col + dim].contiguous()).reshape(-1, dim)
            got = orc.evmap(LEv, LpEv, N, ext, [(colv, 0, dim, prime)])[0]
            assert [int(v) for v in got] == [int(v) for v in ev[i]], f"evaluation {i} (dim {dim}, prime {prime}) differs from the oracle"
        # and the purpose of the map, on one base-field entry: evals = P(xi) (or P(w xi)) for the interpolant of the column
        i = next(k for k in sorted(pick) if ents[k][3] == 1)
        t, width, col, dim, prime = ents[i]
        colv = env.to_np(t.view(-1, width)[:, col].contiguous())
        coef = orc.ntt(colv, len(colv), 1, inverse=True).reshape(-1)          # P(shift x) on the extended domain
        assert not coef[N:].any()
        pt = m3(m3(xi, w_n) if prime else xi, sinv)
        acc = np.zeros(3, np.uint64)
        # Horner in Fp^3 over 2^nBits base-field coefficients, blocked through numpy-free oracle calls would take minutes:
        # evaluate with the power table instead: sum_i c_i pt^i = dot(c, geometric(pt))
        pw = orc.geometric(N, 3, [1, 0, 0], pt)
        got = orc.evmap(pw, pw, N, 0, [(coef[:N].reshape(-1, 1), 0, 1, False)])[0]
        assert [int(v) for v in got] == [int(v) for v in ev[i]], "evaluation is not the interpolant's value at the challenge"
        self.report["evals_checked"] = len(pick)

    def fri(self, data):
        """data: dict from the tool -- roots, queries, opened rows + paths of every tree, FRI challenges, final polynomial"""
        env, orc = self.env, self.orc
        steps, ys = data["steps"], data["ys"]
        # --- the five step-0 trees
        for name, root, width, proofs in data["s0"]:
            for q, y in enumerate(ys):
                p = proofs[q]
                assert [int(v) for v in orc.merkle_root_from_path(p[:width], p[width:], int(y))] == [int(v) for v in root], (name, q)
                self.report["merkle_paths"] += 1
        # --- FRI trees: paths, first tree vs f, folds
        idx = [int(y) for y in ys]
        sinv = orc.inv(7)
        f_rows = data["f_at_queries"]                         # f_2ns[j * 2^nBits1 + (y mod 2^nBits1)] for every j: the group tree 1 opens
        for si in range(1, len(steps)):
            root, width, ng, proofs = data["fri_trees"][si - 1]
            idx = [y % ng for y in idx]
            red = steps[si - 1] - steps[si]
            # shift^-1 after the folds before step si (friProve.cpp:142-148)
            if si >= 2:
                for _ in range(steps[si - 2] - steps[si - 1]):
                    sinv = orc.mul(sinv, sinv)
            for q, y in enumerate(idx):
                p = proofs[q]
                assert [int(v) for v in orc.merkle_root_from_path(p[:width], p[width:], y)] == [int(v) for v in root], ("fri", si, q)
                self.report["merkle_paths"] += 1
                if si == 1:
                    assert (p[:width] == f_rows[q]).all(), "first FRI tree does not open the FRI polynomial"
                    self.report["fri_first_tree_vs_f"] += 1
                # fold of this group at special_x[si] -> entry of the next tree's row / final polynomial
                x = data["special_x"][si]
                got = orc.fri_fold_one(p[:width], steps[si - 1], red, y, x, sinv)
                if si < len(steps) - 1:
                    ng2 = 1 << steps[si + 1]
                    nxt_row = data["fri_trees"][si][3][q]
                    j = y >> steps[si + 1]
                    exp = nxt_row[3 * j:3 * j + 3]
                else:
                    exp = data["final_pol"][3 * y:3 * y + 3]
                assert [int(v) for v in got] == [int(v) for v in exp], ("fold", si, q)
                self.report["fri_folds"] += 1
        # --- every committed section is a low-degree extension, and so is the FRI polynomial: the INTT (HIP kernel, itself checked
        #     against the oracle in tests/test_gpu_parity.py) of sampled columns over the extended domain vanishes above N
        import torch
        capi = env.capi
        NE, N = env.NE, env.N
        col = torch.empty(NE, dtype=torch.int64, device="cuda"); co_ = torch.empty_like(col)
        self.report["low_degree_columns"] = 0
        for name, w, ext in env.secs:
            if not (ext and w) or name == "q_2ns":
                continue
            t = env.sec(name).view(NE, w)
            cols = sorted({0, w - 1} | {int(c) for c in self.rng.integers(0, w, 4)})
            for c in cols:
                col.copy_(t[:, c]); capi.dev_ntt(co_, col, NE, 1, inverse=True)
                assert not bool(co_[N:].any().item()), f"column {c} of {name} is not a polynomial of degree < N on the extended domain"
                self.report["low_degree_columns"] += 1
        # --- final polynomial: degree < size / blow-up
        fp = np.asarray(data["final_pol"], np.uint64).reshape(-1, 3)
        co = orc.ntt(fp, fp.shape[0], 3, inverse=True)
        assert not co[fp.shape[0] >> env.ext:].any(), "final polynomial has too high a degree"
        # --- transcript replay on the oracle
        tr = orc.Transcript()
        tr.put(data["verkey"]); tr.put(data["publics"])
        ch = {}
        tr.put(data["roots"][0]); ch[0] = tr.get_field(); ch[1] = tr.get_field()
        tr.put(data["roots"][1]); ch[2] = tr.get_field(); ch[3] = tr.get_field()
        tr.put(data["roots"][2]); ch[4] = tr.get_field()
        tr.put(data["roots"][3]); ch[7] = tr.get_field()
        tr.put(data["evals"]); ch[5] = tr.get_field(); ch[6] = tr.get_field()
        order = {0: (0, 3), 1: (3, 6), 2: (6, 9), 3: (9, 12), 4: (12, 15), 5: (15, 18), 6: (18, 21), 7: (21, 24)}
        for k, (a, b) in order.items():
            assert [int(v) for v in ch[k]] == [int(v) for v in data["challenges"][a:b]], f"challenge {k}"
        for si in range(len(steps)):
            sx = tr.get_field()
            assert [int(v) for v in sx] == [int(v) for v in data["special_x"][si]], f"special_x {si}"
            if si < len(steps) - 1:
                tr.put(data["fri_trees"][si][0])
            else:
                tr.put(data["final_pol"])
        assert [int(v) for v in tr.get_permutations(len(ys), steps[0])] == [int(v) for v in ys]
        self.report["transcript_replayed"] = True

    # ------------------------------------------------------------------ helpers
    def _program_file(self, step):
        import os
        return os.path.join(R.REF_DIR, f"{self.env.circ}.{step}.xlpexpr")

    def _sections(self, base):
        """(name, width) of the sections a base-domain / extended-domain program may touch, slab order"""
        env = self.env
        out = []
        for name, w, ext in env.secs:
            if w and bool(ext) != base:
                out.append((name, w))
        return out

    def _gather(self, win, base):
        env = self.env
        idx = env.index(win)
        snap = {}
        for name, w in self._sections(base):
            snap[name] = env.to_np(env.sec(name).view(-1, w)[idx]).reshape(len(win), w)
        if base:
            snap["const"] = env.to_np(env.const_n.view(-1, env.nconst)[idx])
            snap["x"] = env.to_np(env.x_n[idx])
        else:
            snap["const"] = env.to_np(env.const_2ns.view(-1, env.nconst)[idx])
            snap["x"] = env.to_np(env.x_2ns[idx])
            snap["zi"] = np.array([env.zi_host[r % len(env.zi_host)] for r in win], np.uint64)
            snap["xdx"] = env.to_np(env.xdx.view(-1, 3)[idx]); snap["xdwx"] = env.to_np(env.xdwx.view(-1, 3)[idx])
        return snap

    def _oracle_buffers(self, snap, secs, total, base):
        env, K = self.env, self.K
        pols = np.zeros(total + 4, np.uint64)
        o = 0
        for name, w in secs:
            pols[o:o + w * K] = snap[name].reshape(-1); o += w * K
        one = np.zeros(1, np.uint64)
        bufs = {"pols": pols, "challenges": np.ascontiguousarray(env.challenges, np.uint64).copy(), "public_inputs": np.ascontiguousarray(env.publics, np.uint64).copy(),
                "evals": env.to_np(env.evals_d).copy()}
        if base:
            bufs.update(const_pols=np.ascontiguousarray(snap["const"]).reshape(-1).copy(), x_n=snap["x"].copy(), const_pols_2ns=one, x_2ns=one, zi=one,
                        x_div_x_sub_xi=np.zeros(3, np.uint64), x_div_x_sub_wxi=np.zeros(3, np.uint64), q_2ns=np.zeros(3, np.uint64), f_2ns=np.zeros(3, np.uint64))
        else:
            qo = [o_ for o_ in self._mini_offsets(secs) if o_[0] == "q_2ns"][0][1]; fo = [o_ for o_ in self._mini_offsets(secs) if o_[0] == "f_2ns"][0][1]
            # q_2ns / f_2ns are sections of the slab AND separate members of StepsParams (steps.hpp:4-19): give the oracle views of the
            # window slab, so that a store through either name lands in the rows that are compared
            bufs.update(const_pols=one, x_n=one, const_pols_2ns=np.ascontiguousarray(snap["const"]).reshape(-1).copy(), x_2ns=snap["x"].copy(), zi=snap["zi"].copy(),
                        x_div_x_sub_xi=np.ascontiguousarray(snap["xdx"]).reshape(-1).copy(), x_div_x_sub_wxi=np.ascontiguousarray(snap["xdwx"]).reshape(-1).copy(),
                        q_2ns=pols[qo:qo + 3 * K], f_2ns=pols[fo:fo + 3 * K])
        return bufs

    def _mini_offsets(self, secs):
        o, out = 0, []
        for name, w in secs:
            out.append((name, o)); o += w * self.K
        return out

    def _relocate(self, code, mini):
        """the as-converted program (real slab addressing of its STARK) -> the K-row window slab `mini`"""
        env = self.env
        real = G.layout(G.CIRCUITS[env.circ]["log_n"], False, env.circ)
        code = code.copy()
        big = 1 << 40

        def move(off, stride):
            for name, (o, w, rows) in real.items():
                if w and w == stride and o <= off < o + w * rows and (off - o) < w:
                    if name not in mini:
                        raise AssertionError(f"program of this domain touches section {name}")
                    return mini[name][0] + (off - o)
            raise AssertionError(f"operand {off} / stride {stride} is in no section")
        for _, _, kind, dim, q in R.operands(code):
            if kind == E.K_POLS:
                code[q] = move(int(code[q]), int(code[q + 1]))
            elif kind == E.K_POLS_PRIME:
                code[q] = move(int(code[q]), int(code[q + 3])); code[q + 2] = big
            elif kind in (E.K_CONST_PRIME, E.K_CONST2NS_PRIME):
                code[q + 2] = big
        return code
