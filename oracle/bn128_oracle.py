"""CPU restatement (plain Python integers) of the BN128 side of StarkRecursiveF -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path never does.
Each function cites the reference lines it follows (paths relative to /root/reference):
  Poseidon over the BN254 scalar field, circomlib's optimized form   src/poseidon_opt/poseidon_opt.cpp:3-100
  MerkleTreeBN128 (16-ary, linear hash of 3 Goldilocks per Fr)       src/starkpil/merkleTree/merkleTreeBN128.cpp:62-260
  TranscriptBN128                                                     src/starkpil/transcript/transcriptBN128.cpp:5-141
The field library (ffiasm `RawFr`, generated from src/ffiasm/fr.asm) is standard 4 x 64-bit Montgomery arithmetic with R = 2^256;
elements are kept here as ordinary integers mod r and converted to / from Montgomery limbs only at the byte boundary.

Pinning: the reference snapshot holds no BN128 golden data (no recursivef proof, no BN128 tree).  What it does hold pins the ELEMENT
REPRESENTATION: the field constants of src/ffiasm/fr.asm / fq.asm (q, R^2, R^3, np -> tests/golden/bn254_field_constants.json) fix RawFr / RawFq as
4 x 64-bit Montgomery residues with R = 2^256, and the four G1 commitments of testvectors/finalProof/proof.json fix the curve.  The permutation is
pinned by circomlib's published known answers for t = 3 and t = 5 (tests/test_bn128.py); the 16-ary tree layout, the leaf rule and the
transcript are restatement-only -- "parity unpinned" for those, stated in DESIGN.md (a recursivef proof or .consttree would pin them).
"""
import struct

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # BN254 scalar field
MONT_R = (1 << 256) % R_MOD
MONT_RINV = pow(MONT_R, -1, R_MOD)
N_ROUNDS_F = 8
N_ROUNDS_P = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]           # poseidon_opt.hpp:18
ARITY = 16


def to_mont_limbs(x):
    v = (x % R_MOD) * MONT_R % R_MOD
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_mont_limbs(l):
    v = sum(int(l[i]) << (64 * i) for i in range(4))
    return v * MONT_RINV % R_MOD


class Constants:
    """tables of xlayer_prover_amd/chelpers/poseidon_bn128.bin (tools/gen_poseidon_bn128.py), as integers mod r"""

    def __init__(self, path):
        raw = open(path, "rb").read()
        assert raw[:8] == b"XLPBN128"
        counts = [struct.unpack_from("<4Q", raw, 16 + 32 * k) for k in range(16)]
        off = 16 + 32 * 16
        self.C, self.S, self.M, self.P = [], [], [], []
        for k in range(16):
            tabs = []
            for n in counts[k]:
                vals = []
                for _ in range(n):
                    vals.append(from_mont_limbs(struct.unpack_from("<4Q", raw, off)))
                    off += 32
                tabs.append(vals)
            t = k + 2
            self.C.append(tabs[0]); self.S.append(tabs[1])
            self.M.append([tabs[2][j * t:(j + 1) * t] for j in range(t)])
            self.P.append([tabs[3][j * t:(j + 1) * t] for j in range(t)])


def poseidon(K, state):
    """Poseidon_opt::hash(vector<FrElement>&) (poseidon_opt.cpp:9-66): permutes a state of 2..17 elements in place order"""
    t = len(state)
    assert 2 <= t <= 17
    nP = N_ROUNDS_P[t - 2]
    c, s, m, p = K.C[t - 2], K.S[t - 2], K.M[t - 2], K.P[t - 2]
    st = [x % R_MOD for x in state]

    def mix(mat, v):                                              # poseidon_opt.cpp:89-100: new[i] = sum_j mat[j][i] * v[j]
        return [sum(mat[j][i] * v[j] for j in range(t)) % R_MOD for i in range(t)]

    st = [(st[i] + c[i]) % R_MOD for i in range(t)]
    for r in range(N_ROUNDS_F // 2 - 1):
        st = [(pow(st[i], 5, R_MOD) + c[(r + 1) * t + i]) % R_MOD for i in range(t)]
        st = mix(m, st)
    st = [(pow(st[i], 5, R_MOD) + c[(N_ROUNDS_F // 2) * t + i]) % R_MOD for i in range(t)]
    st = mix(p, st)
    for r in range(nP):
        st[0] = (pow(st[0], 5, R_MOD) + c[(N_ROUNDS_F // 2 + 1) * t + r]) % R_MOD
        base = (2 * t - 1) * r
        s0 = sum(s[base + j] * st[j] for j in range(t)) % R_MOD
        for j in range(1, t):
            st[j] = (st[j] + st[0] * s[base + t + j - 1]) % R_MOD
        st[0] = s0
    for r in range(N_ROUNDS_F // 2 - 1):
        st = [(pow(st[i], 5, R_MOD) + c[(N_ROUNDS_F // 2 + 1) * t + nP + r * t + i]) % R_MOD for i in range(t)]
        st = mix(m, st)
    st = [pow(x, 5, R_MOD) for x in st]
    return mix(m, st)


def get_num_nodes(n):
    """MerkleTreeBN128::getNumNodes (merkleTreeBN128.cpp:62-84)"""
    n_tmp = n
    next_n = (n_tmp - 1) // ARITY + 1
    acc = next_n * ARITY
    while n_tmp > 1:
        n_tmp = next_n
        next_n = (n_tmp - 1) // ARITY + 1
        acc += next_n * ARITY if n_tmp > 1 else 1
    return acc


def pack_row(row):
    """3 Goldilocks elements per Fr, little-endian 64-bit limbs (merkleTreeBN128.cpp:97-113)"""
    out = []
    for j in range(0, len(row), 3):
        out.append(sum(int(x) << (64 * k) for k, x in enumerate(row[j:j + 3])))
    return out


def linear_hash_row(K, row):
    """one leaf of MerkleTreeBN128::linearHash (merkleTreeBN128.cpp:89-157)"""
    w = len(row)
    if w > 4:
        els = pack_row(row)
        node, pending = 0, len(els)
        while pending > 0:
            take = min(pending, 16)
            chunk = els[len(els) - pending:len(els) - pending + take]
            if take == 16:
                node = poseidon(K, [node] + chunk)[0]
            else:
                node = poseidon(K, [node] + chunk)[0]            # t = pending + 1 (elements_last)
            pending -= take
        return node
    return sum(int(x) << (64 * k) for k, x in enumerate(row)) % R_MOD   # up to 4 elements in one Fr, no hash (:146-156)


def merkelize(K, src):
    """MerkleTreeBN128::merkelize (merkleTreeBN128.cpp:159-186) -> list of numNodes integers (slots never written stay 0)"""
    h = len(src)
    nodes = [0] * get_num_nodes(h)
    for i in range(h):
        nodes[i] = linear_hash_row(K, src[i])
    cursor, n256 = 0, h
    next_n = (n256 - 1) // 16 + 1
    cursor_next = n256
    while n256 > 1:
        batches = (n256 + 15) // 16
        for i in range(batches):
            num = n256 if batches == 1 else 16
            els = [0] * 17
            for k in range(num):
                # the reference copies numHashes elements from cursor[i * 16 ..]; in the last batch of a multi-batch level they may
                # lie beyond n256 -- inside the level's own padding (levels are allocated in multiples of 16, calloc'ed)
                els[1 + k] = nodes[cursor + i * 16 + k]
            nodes[cursor_next + i] = poseidon(K, els)[0]
        n256 = next_n
        next_n = (n256 - 1) // 16 + 1
        cursor = cursor_next
        cursor_next = cursor + max(n256, 16)
    return nodes


def nodes_to_limbs(nodes):
    import numpy as np
    out = np.zeros((len(nodes), 4), dtype=np.uint64)
    for i, v in enumerate(nodes):
        out[i] = to_mont_limbs(v)
    return out


def merkle_proof(nodes, idx, h):
    """merkle_genMerkleProof (merkleTreeBN128.cpp:248-260): per level the 16 siblings of the group idx belongs to"""
    out, offset, n = [], 0, h
    while n > 1:
        si = idx & ~0xF
        out.append(nodes[offset + si:offset + si + 16])
        next_n = (n - 1) // 16 + 1
        offset += next_n * 16
        idx >>= 4
        n = next_n
    return out


class TranscriptBN128:
    """transcriptBN128.cpp:5-141"""

    def __init__(self, K):
        self.K = K
        self.state = [0]
        self.pending = []
        self.out = [0]
        self.out3 = []

    def put(self, xs):                    # Goldilocks elements or Fr integers: both enter as one Fr each (:5-24)
        for x in xs:
            self.pending.append(int(x) % R_MOD)
            self.out = []
            if len(self.pending) == 16:
                self._update()

    def _update(self):
        while len(self.pending) < 16:
            self.pending.append(0)
        self.out = poseidon(self.K, self.out + self.state + self.pending)
        self.state = [self.out[0]]
        self.out3 = []
        self.pending = []

    def get_fields253(self):
        if self.out:
            return self.out.pop(0)
        self._update()
        return self.get_fields253()

    def get_fields1(self):
        if self.out3:
            return self.out3.pop(0)
        if self.out:
            v = self.out.pop(0)
            self.out3 += [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(3)]
            return self.get_fields1()
        self._update()
        return self.get_fields1()

    def get_field(self):
        return [self.get_fields1() for _ in range(3)]

    def get_permutations(self, n, n_bits):
        total = n * n_bits
        nf = (total - 1) // 253 + 1
        fields = [self.get_fields253() for _ in range(nf)]
        res, cf, cb = [], 0, 0
        for _ in range(n):
            a = 0
            for j in range(n_bits):
                if (fields[cf] >> cb) & 1:
                    a += 1 << j
                cb += 1
                if cb == 253:
                    cb, cf = 0, cf + 1
            res.append(a)
        return res


# ------------------------------------------------------------------------------------------------ FFT over Fr (fflonk prover)
def fft_root(logn):
    """w_n of FFT<RawFr> (src/ffiasm/fft.c.hpp:52-84): nqr^((q - 1) / n) with nqr = 5, the smallest quadratic non-residue"""
    assert pow(2, (R_MOD - 1) // 2, R_MOD) == 1 and pow(3, (R_MOD - 1) // 2, R_MOD) == 1 and pow(4, (R_MOD - 1) // 2, R_MOD) == 1
    assert pow(5, (R_MOD - 1) // 2, R_MOD) == R_MOD - 1
    return pow(5, (R_MOD - 1) >> logn, R_MOD)


def fft(a):
    """FFT<Field>::fft (fft.c.hpp:181-202): bit reversal, then log n radix-2 stages; natural order in and out"""
    n = len(a)
    logn = n.bit_length() - 1
    assert 1 << logn == n
    w = fft_root(logn)
    a = [a[int(format(i, "0%db" % logn)[::-1], 2)] if logn else a[i] for i in range(n)]
    for s in range(1, logn + 1):
        m, half = 1 << s, 1 << (s - 1)
        ws = pow(w, n >> s, R_MOD)
        for k in range(0, n, m):
            x = 1
            for j in range(half):
                t = x * a[k + j + half] % R_MOD
                u = a[k + j]
                a[k + j] = (u + t) % R_MOD
                a[k + j + half] = (u - t) % R_MOD
                x = x * ws % R_MOD
    return a


def ifft(a):
    """FFT<Field>::ifft (fft.c.hpp:205-221)"""
    n = len(a)
    b = fft(a)
    ninv = pow(n, -1, R_MOD)
    return [b[(n - i) % n] * ninv % R_MOD for i in range(n)]


# ------------------------------------------------------------------------------------------------ G1 multi-scalar multiplication
# ParallelMultiexp<Curve>::multiexp (src/ffiasm/multiexp.c.hpp:118-164) computes sum_i scalars[i] * bases[i] on alt_bn128 G1
# (y^2 = x^3 + 3 over Fq, src/ffiasm/alt_bn128.cpp); restated here as plain double-and-add per point (a different algorithm from
# both the reference's and the GPU's bucket methods).  Points are (x, y) integers, None = infinity.
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
Q_MONT_R = (1 << 256) % Q_MOD
Q_MONT_RINV = pow(Q_MONT_R, -1, Q_MOD)
G1 = (1, 2)


def g1_on_curve(p):
    return p is None or (p[1] * p[1] - (p[0] ** 3 + 3)) % Q_MOD == 0


def _jdbl(p):
    X, Y, Z = p
    if Z == 0:
        return p
    A, B = X * X % Q_MOD, Y * Y % Q_MOD
    C = B * B % Q_MOD
    D = 2 * ((X + B) ** 2 - A - C) % Q_MOD
    E = 3 * A % Q_MOD
    X3 = (E * E - 2 * D) % Q_MOD
    return X3, (E * (D - X3) - 8 * C) % Q_MOD, 2 * Y * Z % Q_MOD


def _jadd(p, q):
    if q[2] == 0:
        return p
    if p[2] == 0:
        return q
    Z1Z1, Z2Z2 = p[2] * p[2] % Q_MOD, q[2] * q[2] % Q_MOD
    U1, U2 = p[0] * Z2Z2 % Q_MOD, q[0] * Z1Z1 % Q_MOD
    S1, S2 = p[1] * q[2] * Z2Z2 % Q_MOD, q[1] * p[2] * Z1Z1 % Q_MOD
    if U1 == U2:
        return _jdbl(p) if S1 == S2 else (1, 1, 0)
    H, r = (U2 - U1) % Q_MOD, (S2 - S1) % Q_MOD
    HH = H * H % Q_MOD
    HHH = H * HH % Q_MOD
    V = U1 * HH % Q_MOD
    X3 = (r * r - HHH - 2 * V) % Q_MOD
    return X3, (r * (V - X3) - S1 * HHH) % Q_MOD, p[2] * q[2] * H % Q_MOD


def _to_affine(p):
    if p[2] == 0:
        return None
    zi = pow(p[2], -1, Q_MOD)
    return p[0] * zi * zi % Q_MOD, p[1] * zi * zi * zi % Q_MOD


def g1_mul(p, k):
    if p is None or k == 0:
        return None
    acc, base = (1, 1, 0), (p[0], p[1], 1)
    for b in range(k.bit_length() - 1, -1, -1):
        acc = _jdbl(acc)
        if (k >> b) & 1:
            acc = _jadd(acc, base)
    return _to_affine(acc)


def g1_add(p, q):
    a = (1, 1, 0) if p is None else (p[0], p[1], 1)
    b = (1, 1, 0) if q is None else (q[0], q[1], 1)
    return _to_affine(_jadd(a, b))


def msm(bases, scalars):
    """sum_i scalars[i] * bases[i]; scalars are raw integers < 2^256 (not reduced, as getChunk reads them, multiexp.c.hpp:23-35)"""
    acc = (1, 1, 0)
    for p, k in zip(bases, scalars):
        if p is None or k == 0:
            continue
        q = g1_mul(p, k)
        if q is not None:
            acc = _jadd(acc, (q[0], q[1], 1))
    return _to_affine(acc)


def g1_to_limbs(p):
    """affine point -> 8 uint64: x, y as Fq Montgomery limbs ((0, 0) = infinity, the reference's zeroAffine)"""
    if p is None:
        return [0] * 8
    out = []
    for c in p:
        v = c * Q_MONT_R % Q_MOD
        out += [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    return out


def g1_from_limbs(l):
    x = sum(int(l[i]) << (64 * i) for i in range(4)) * Q_MONT_RINV % Q_MOD
    y = sum(int(l[4 + i]) << (64 * i) for i in range(4)) * Q_MONT_RINV % Q_MOD
    return None if x == 0 and y == 0 else (x, y)
