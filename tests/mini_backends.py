"""Backends of the toy prover (tests/mini_genproof_flow.py): the HIP library through its C ABI, and the CPU oracle."""
import numpy as np

from tests import mini_genproof_flow as flow

P = flow.P
W32 = 7277203076849721926


def gl_w(n):
    w = W32
    for _ in range(32 - n):
        w = w * w % P
    return w


def _merkle_proof_host(src, nodes, w, h, idx):
    row = [int(v) for v in src[idx * w:(idx + 1) * w]]
    sib, off, cnt, i = [], 0, h, idx
    while cnt > 1:
        s = i ^ 1
        sib.append([int(v) for v in nodes[off + 4 * s: off + 4 * s + 4]])
        off += 4 * cnt; cnt //= 2; i //= 2
    return row, sib


class OracleBackend:
    def __init__(self, orc): self.o = orc
    def sync(self): pass
    def w(self, n): return gl_w(n)
    def zeros(self, n): return np.zeros(n, np.uint64)
    def from_host(self, a): return np.ascontiguousarray(a, dtype=np.uint64).copy()
    def to_host(self, a): return a
    def copy_into(self, dst, src): dst[:] = src
    def extend(self, dst, src, ne, n, w): dst[:] = self.o.extend_pol(src, ne, n, w).reshape(-1)
    def ntt(self, dst, src, n, w, inverse): dst[:] = self.o.ntt(src.copy(), n, w, inverse).reshape(-1)
    def merkle(self, src, w, h): return self.o.merkletree(src, w, h)
    def root(self, nodes, h): return [int(v) for v in nodes[-4:]]
    def get_cols(self, sec, rows, W, c, d): return np.ascontiguousarray(sec.reshape(rows, W)[:, c:c + d]).reshape(-1)
    def set_cols(self, sec, rows, W, c, d, vals): sec.reshape(rows, W)[:, c:c + d] = vals.reshape(rows, d)
    def h1h2(self, h1, h2, f, t, n, dim):
        a, b = self.o.calculate_h1h2(f, t, dim); h1[:] = a.reshape(-1); h2[:] = b.reshape(-1)
    def calc_z(self, z, num, den, n): z[:] = self.o.calculate_z(num, den).reshape(-1)
    def expr_run(self, prog, bufs, nc, nc2, row0, nrows): self.o.expr_run(prog.code, prog.ntmp1, prog.ntmp3, bufs, row0, nrows, nc, nc2)
    def geometric(self, n, dim, first, ratio): return self.o.geometric(n, dim, np.array(first, np.uint64), np.array(ratio, np.uint64)).reshape(-1).copy()
    def x_div(self, x, xi, n): return self.o.x_div_x_sub_xi(x, np.array(xi, np.uint64)).reshape(-1)
    def q_split(self, qq1, n, ne, qdeg, qdim, s): return self.o.q_split(qq1, n, ne, qdeg, qdim, s)
    def evmap(self, LEv, LpEv, n, eb, ents):
        return self.o.evmap(LEv, LpEv, n, eb, [(sec.reshape(-1, w), c, d, pr) for sec, w, c, d, pr in ents])
    def fri_fold(self, pol, bits, red, x, sinv): return self.o.fri_fold(pol, bits, red, np.array(x, np.uint64), sinv).reshape(-1)
    def transpose3(self, pol2, n, tbits): return self.o.transpose3(pol2, n, tbits).reshape(-1)
    def perm(self, st): return self.o.perm(np.array(st, np.uint64))
    def proof(self, src, nodes, w, h, idx): return _merkle_proof_host(src, nodes, w, h, idx)
    def to_torch(self, a):
        import torch
        return torch.from_numpy(a.view(np.int64))
    def from_torch(self, t): return t.contiguous().numpy().view(np.uint64)


class GpuBackend:
    def __init__(self, xlp):
        import torch
        self.x, self.t, self.progs = xlp, torch, {}
    def sync(self): self.x.sync()
    def w(self, n): return gl_w(n)
    def zeros(self, n): return self.t.zeros(n, dtype=self.t.int64, device="cuda")
    def from_host(self, a): return self.x.from_np(np.ascontiguousarray(a, dtype=np.uint64))
    def to_host(self, a): return self.x.to_np(a)
    def copy_into(self, dst, src): dst.copy_(src)
    def extend(self, dst, src, ne, n, w): self.x.dev_extend(dst, src, ne, n, w)
    def ntt(self, dst, src, n, w, inverse): self.x.dev_ntt(dst, src, n, w, inverse)
    def merkle(self, src, w, h):
        nodes = self.zeros((2 * h - 1) * 4); self.x.dev_merkle(nodes, src, w, h); return nodes
    def root(self, nodes, h): return [int(v) for v in self.x.to_np(nodes[-4:])]
    def get_cols(self, sec, rows, W, c, d): return sec.view(rows, W)[:, c:c + d].contiguous().view(-1)
    def set_cols(self, sec, rows, W, c, d, vals): sec.view(rows, W)[:, c:c + d] = vals.view(rows, d)
    def h1h2(self, h1, h2, f, t, n, dim): self.x.dev_calculate_h1h2(h1, h2, f, t, n, dim)
    def calc_z(self, z, num, den, n): self.x.dev_calculate_z(z, num, den, n)
    def expr_run(self, prog, bufs, nc, nc2, row0, nrows):
        if id(prog) not in self.progs or self.progs[id(prog)][0] is not prog:
            self.progs[id(prog)] = (prog, self.x.ExprProgram(prog.code, prog.ntmp1, prog.ntmp3))
        self.progs[id(prog)][1].run(self.x.steps_params(n_const_pols=nc, n_const_pols_2ns=nc2, **bufs), row0, nrows)
    def geometric(self, n, dim, first, ratio):
        out = self.zeros(n * dim); self.x.dev_geometric(out, n, dim, np.array(first, np.uint64), np.array(ratio, np.uint64)); return out
    def x_div(self, x, xi, n):
        out = self.zeros(3 * n); self.x.dev_x_div_x_sub_xi(out, x, np.array(xi, np.uint64), n); return out
    def q_split(self, qq1, n, ne, qdeg, qdim, s):
        out = self.zeros(ne * qdeg * qdim); self.x.dev_q_split(out, qq1, n, ne, qdeg, qdim, s); return out
    def evmap(self, LEv, LpEv, n, eb, ents):
        ev = self.zeros(3 * len(ents)); self.x.dev_evmap(ev, LEv, LpEv, n, eb, ents); return self.x.to_np(ev).reshape(-1, 3)
    def fri_fold(self, pol, bits, red, x, sinv):
        out = self.zeros(3 << (bits - red)); self.x.dev_fri_fold(out, pol, bits, red, np.array(x, np.uint64), sinv); return out
    def transpose3(self, pol2, n, tbits):
        out = self.zeros(3 * n); self.x.dev_transpose3(out, pol2, n, tbits); return out
    def perm(self, st): return self.x.perm(np.array(st, np.uint64))
    def to_torch(self, a): return a
    def from_torch(self, t): return t.contiguous()
    def proof(self, src, nodes, w, h, idx):
        # MerkleTreeGL::getGroupProof (merkleTreeGL.cpp:12-35): only the row and log2(h) sibling nodes leave the GPU
        row = [int(v) for v in self.x.to_np(src[idx * w:(idx + 1) * w])]
        pos, off, cnt, i = [], 0, h, idx
        while cnt > 1:
            pos.append(off + 4 * (i ^ 1)); off += 4 * cnt; cnt //= 2; i //= 2
        if not pos:
            return row, []
        ix = (self.t.tensor(pos, device="cuda").view(-1, 1) + self.t.arange(4, device="cuda").view(1, 4)).view(-1)
        sib = self.x.to_np(nodes[ix]).reshape(-1, 4)
        return row, [[int(v) for v in r] for r in sib]




# ---- StarkRecursiveF shape (starkRecursiveF.cpp:150-175): the same flow with MerkleTreeBN128 trees and TranscriptBN128.
# Roots are one Fr (an integer mod r), openings are the row + the 16-node groups of merkle_genMerkleProof.
BN_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BN_RINV = pow(1 << 256, -1, BN_R)


def _fr_from_limbs(l):
    return sum(int(l[i]) << (64 * i) for i in range(4)) * BN_RINV % BN_R


class OracleBackendBN128(OracleBackend):
    def __init__(self, orc, bo, K):
        super().__init__(orc); self.bo, self.K = bo, K
    def merkle(self, src, w, h): return self.bo.merkelize(self.K, np.asarray(src).reshape(h, w))
    def root(self, nodes, h): return [nodes[-1]]
    def proof(self, src, nodes, w, h, idx):
        return [int(v) for v in np.asarray(src).reshape(h, w)[idx]], [list(g) for g in self.bo.merkle_proof(nodes, idx, h)]
    def transcript(self): return self.bo.TranscriptBN128(self.K)


class GpuBackendBN128(GpuBackend):
    def merkle(self, src, w, h):
        nodes = self.zeros(self.x.merkle_bn128_num_nodes(h) * 4); self.x.dev_merkle_bn128(nodes, src, w, h); return nodes
    def root(self, nodes, h): return [_fr_from_limbs(self.x.to_np(nodes[-4:]))]
    def proof(self, src, nodes, w, h, idx):
        row = [int(v) for v in self.x.to_np(src[idx * w:(idx + 1) * w])]
        groups, off, n, i = [], 0, h, idx
        while n > 1:
            si = i & ~0xF
            g = self.x.to_np(nodes[(off + si) * 4:(off + si + 16) * 4]).reshape(16, 4)
            groups.append([_fr_from_limbs(r) for r in g])
            nn = (n - 1) // 16 + 1
            off += nn * 16; i >>= 4; n = nn
        return row, groups
    def transcript(self):
        from xlayer_prover_amd.transcript import HostTranscriptBN128
        R = (1 << 256) % BN_R
        def perm17(st):
            limbs = np.array([[((x * R % BN_R) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)] for x in st], np.uint64)
            out = self.x.poseidon_bn128(limbs.reshape(1, 17, 4), 17)[0]
            return [_fr_from_limbs(r) for r in out]
        return HostTranscriptBN128(perm17)
