"""CPU restatement (plain Python integers) of the fflonk prover of the final proof -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path never does.
What it follows (paths relative to /root/reference):
  FflonkProver::prove and its five rounds          src/rapidsnark/fflonk_prover.c.hpp:437-1716   (called at src/prover/prover.cpp:931)
  Polynomial / Evaluations / CPolynomial           src/rapidsnark/polynomial/{polynomial,evaluations,cpolynomial}.c.hpp
  Keccak256Transcript                              src/rapidsnark/keccak_256_transcript.c.hpp:10-78 (Keccak[r=1088, c=512], suffix 0x01: keccak_wrapper.cpp:9)
  the zkey / wtns containers                       src/rapidsnark/binfile_utils.cpp:15-56, zkey_fflonk.cpp:18-49 (sections: zkey_fflonk.hpp:9-25), wtns_utils.cpp:13-26
  the proof's JSON form                            src/rapidsnark/snark_proof.c.hpp:62-88

Field elements are ordinary integers mod r here; Montgomery limbs (oracle/bn128_oracle.py) only at the byte boundary.

PINNING.  The reference snapshot holds no fflonk zkey, no witness and no fflonk known answer (testvectors/finalProof/proof.json is a proof without
the key it was made with): the prover is "parity unpinned" against the reference.  What stands in for a reference run:
  * `setup()` builds a SYNTHETIC proving key for a random PLONK circuit in the byte format the reference reads (section by section; the setup
    itself is snarkjs' and not in the snapshot: C0 = QL, QR, QO, QM, QC, S1, S2, S3 interleaved by 8, the last two rows of the domain left out
    of the copy constraints because the prover blinds there, fflonk_prover.c.hpp:708-709) from a KNOWN tau;
  * `verify()` is the fflonk verification equation (the fflonk paper, Gabizon and Williamson 2021, as the prover's rounds imply it; the batched opening check of the three combined
    polynomials at the roots of xi) with the pairing replaced by the trapdoor:  e(A, [1]_2) = e(W2, [tau]_2)  <=>  A = tau * W2 in G1.
    A proof only passes if every round of the prover -- transcript, quotients, interpolations, openings -- is what the protocol says.
  * the SHAPE of a proof -- four commitments, sixteen evaluations -- is that of the reference's own fflonk proof (testvectors/finalProof/proof.json,
    24 words: tests/golden/fflonk_proof_g1_points.json); its values cannot be re-derived without its verification key;
  * Keccak-256: the sponge with the NIST suffix equals hashlib.sha3_256 on one- and many-block messages (permutation, absorption and
    padding pinned by an independent implementation); the 0x01 suffix by the digests of "", "abc" and the fox sentence (tests/test_fflonk.py).
"""
import os
import struct

from .bn128_oracle import (G1, Q_MOD, R_MOD, fft_root, from_mont_limbs, g1_add, g1_from_limbs, g1_mul, g1_to_limbs, to_mont_limbs)

P = R_MOD
FFLONK_PROTOCOL_ID = 10                       # zkey.hpp:9
SEC = {"header": 2, "additions": 3, "amap": 4, "bmap": 5, "cmap": 6, "ql": 7, "qr": 8, "qm": 9, "qo": 10, "qc": 11,
       "s1": 12, "s2": 13, "s3": 14, "lagrange": 15, "ptau": 16, "c0": 17}      # zkey_fflonk.hpp:9-25


# ------------------------------------------------------------------------------------------------ Keccak-256
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001, 0x8000000080008081,
       0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B,
       0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A, 0x8000000080008081,
       0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _keccak_f(A):
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & _M64 if n else v
    for rc in _RC:
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = rol(A[x][y], _ROT[x][y])
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= rc
    return A


def keccak256(data, suffix=0x01):
    """Keccak(1088, 512, data, 0x01, 32 bytes): the original Keccak padding (keccak_wrapper.cpp:9), Ethereum's keccak256.
    suffix = 0x06 gives SHA3-256: the tests compare that form with hashlib, which pins the permutation and the sponge independently"""
    rate = 136
    msg = bytearray(data) + bytes([suffix])
    msg += b"\x00" * ((-len(msg)) % rate)
    msg[-1] |= 0x80
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            A[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        A = _keccak_f(A)
    return b"".join(A[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


class Transcript:
    """Keccak256Transcript: scalars as 32 big-endian bytes of the value, points as x || y (the point at infinity adds nothing,
    keccak_256_transcript.c.hpp:55-66); the challenge = the digest read as a big-endian integer, mod r (fr.cpp:291-295,314-327)"""

    def __init__(self):
        self.data = b""

    def reset(self):
        self.data = b""

    def add_scalar(self, v):
        self.data += int(v % P).to_bytes(32, "big")

    def add_point(self, pt):
        if pt is not None:
            self.data += int(pt[0]).to_bytes(32, "big") + int(pt[1]).to_bytes(32, "big")

    def challenge(self):
        return int.from_bytes(keccak256(self.data), "big") % P


# ------------------------------------------------------------------------------------------------ field helpers
def inv(a):
    return pow(a % P, -1, P)


def fft(a):
    """FFT<RawFr>::fft (src/ffiasm/fft.c.hpp:181-202): natural order in and out, w = 5^((r - 1) / n)"""
    n = len(a)
    logn = n.bit_length() - 1
    assert 1 << logn == n
    a = list(a)
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    w = fft_root(logn) if logn else 1
    for s in range(1, logn + 1):
        m, half = 1 << s, 1 << (s - 1)
        ws = pow(w, n >> s, P)
        tw = [1] * half
        for k in range(1, half):
            tw[k] = tw[k - 1] * ws % P
        for k in range(0, n, m):
            for jj in range(half):
                t = tw[jj] * a[k + jj + half] % P
                u = a[k + jj]
                a[k + jj] = (u + t) % P
                a[k + jj + half] = (u - t) % P
    return a


def ifft(a):
    n = len(a)
    b = fft(a)
    ninv = inv(n)
    return [b[(n - i) % n] * ninv % P for i in range(n)]


def poly_eval(c, x):
    """Polynomial::fastEvaluate (polynomial.c.hpp:184-214): the value; the reference splits Horner over threads"""
    r = 0
    for v in reversed(c):
        r = (r * x + v) % P
    return r


def degree(c):
    d = len(c) - 1
    while d > 0 and c[d] % P == 0:
        d -= 1
    return d


def div_zh(c, n, extension=4):
    """Polynomial::divZh (polynomial.c.hpp:537-569): coefficients of c / (X^n - 1), in place over chunks of n"""
    c = list(c)
    for i in range(n):
        c[i] = (-c[i]) % P
    chunks = len(c) // n
    for i in range(chunks - 1):
        for j in range(n):
            c[(i + 1) * n + j] = (c[i * n + j] - c[(i + 1) * n + j]) % P
            if i > n * (extension - 1) - extension and c[(i + 1) * n + j]:
                raise ValueError("Polynomial is not divisible")
    return c


def div_by_zerofier(c, n, beta):
    """Polynomial::divByZerofier (polynomial.c.hpp:571-634): c / (X^n - beta)"""
    c = list(c)
    deg = degree(c)
    ib = inv(beta)
    for i in range(n):
        c[i] = (-ib) * c[i] % P
    chunks = len(c) // n
    for i in range(chunks - 1):
        for j in range(n):
            i1 = (i + 1) * n + j
            if i1 > deg:
                break
            c[i1] = (c[i * n + j] - c[i1]) * ib % P
            if i > deg - n and c[i1]:
                raise ValueError("Polynomial is not divisible")
    return c


def poly_mul_x_sub(c, v):
    out = [0] * (len(c) + 1)
    for i, a in enumerate(c):
        out[i + 1] = (out[i + 1] + a) % P
        out[i] = (out[i] - a * v) % P
    return out


def zerofier(xs):
    """Polynomial::zerofierPolynomial (polynomial.c.hpp:714-731)"""
    c = [1]
    for x in xs:
        c = poly_mul_x_sub(c, x)
    return c


def lagrange_interpolation(xs, ys):
    """Polynomial::lagrangePolynomialInterpolation (polynomial.c.hpp:674-712)"""
    out = [0] * len(xs)
    for i in range(len(xs)):
        c = [1]
        for j in range(len(xs)):
            if j != i:
                c = poly_mul_x_sub(c, xs[j])
        f = ys[i] * inv(poly_eval(c, xs[i])) % P
        for k in range(len(c)):
            out[k] = (out[k] + c[k] * f) % P
    return out


def interleave(pols, n):
    """CPolynomial::getPolynomial (cpolynomial.c.hpp:39-63): coefficient i of polynomial j at i * n + j"""
    maxdeg = max(degree(p) * n + j for j, p in enumerate(pols))
    length = 2 ** (int.bit_length(maxdeg - 1))                     # 2^(floor(log2(maxdeg - 1)) + 1)
    out = [0] * max(length, maxdeg + 1)
    for j, p in enumerate(pols):
        for i in range(degree(p) + 1):
            out[i * n + j] = p[i]
    return out


# ------------------------------------------------------------------------------------------------ containers
def write_binfile(path, ftype, version, sections):
    """sections: [(id, bytes)] (binfile_utils.cpp:15-56: 4 type bytes, u32 version, u32 count, then per section u32 id, u64 size, data)"""
    with open(path, "wb") as f:
        f.write(ftype.encode()[:4]); f.write(struct.pack("<II", version, len(sections)))
        for sid, data in sections:
            f.write(struct.pack("<IQ", sid, len(data))); f.write(data)


def read_binfile(path, ftype):
    raw = open(path, "rb").read()
    assert raw[:4] == ftype.encode(), raw[:4]
    version, ns = struct.unpack_from("<II", raw, 4)
    pos, out = 12, {}
    for _ in range(ns):
        sid, size = struct.unpack_from("<IQ", raw, pos)
        out[sid] = raw[pos + 12:pos + 12 + size]
        pos += 12 + size
    return out


def fr_mont_bytes(vals):
    return b"".join(struct.pack("<4Q", *to_mont_limbs(v)) for v in vals)


def fr_from_mont_bytes(b):
    return [from_mont_limbs(struct.unpack_from("<4Q", b, 32 * i)) for i in range(len(b) // 32)]


def g1_bytes(p):
    return struct.pack("<8Q", *g1_to_limbs(p))


def g1_from_bytes(b, i=0):
    return g1_from_limbs(struct.unpack_from("<8Q", b, 64 * i))


# ------------------------------------------------------------------------------------------------ a synthetic circuit and its proving key
class Rng:
    def __init__(self, seed):
        self.s = seed & _M64 or 1

    def next(self):
        self.s ^= (self.s << 13) & _M64; self.s ^= self.s >> 7; self.s ^= (self.s << 17) & _M64
        return self.s

    def fr(self):
        return (self.next() | self.next() << 64 | self.next() << 128 | self.next() << 192) % P

    def below(self, n):
        return self.next() % n


def setup(power, n_public=1, n_additions=3, seed=1, tau=None):
    """-> (zkey dict, witness list, trapdoor tau).  A random PLONK circuit on n = 2^power rows: n_public public-input rows, then
    multiplication / addition / constant gates over a pool of signals (so the copy constraints are non-trivial), some signals being
    `additions` (linear combinations the prover computes itself, fflonk_prover.c.hpp:598-612)."""
    n = 1 << power
    rng = Rng(seed)
    tau = tau if tau is not None else rng.fr()
    n_constraints = n - 2 - rng.below(3)                          # the last two rows are the prover's blinding rows
    # signals: 0 unused (the prover zeroes it), 1..n_public publics, then free signals, the last n_additions computed by the prover
    pool = [0] + [rng.fr() for _ in range(n_public)]
    n_free = max(8, n // 3)
    pool += [rng.fr() for _ in range(n_free)]
    additions = []
    base = len(pool)
    for _ in range(n_additions):
        s1, s2 = 1 + rng.below(base - 1), 1 + rng.below(base - 1)
        f1, f2 = rng.fr(), rng.fr()
        additions.append((s1, s2, f1, f2))
        pool.append((f1 * pool[s1] + f2 * pool[s2]) % P)
    n_vars = len(pool)
    A, B, C = [0] * n_constraints, [0] * n_constraints, [0] * n_constraints
    ql, qr, qm, qo, qc = ([0] * n for _ in range(5))
    for i in range(n_constraints):
        if i < n_public:                                          # a - public_i = 0 (PI(X) = -sum public_i L_i(X))
            A[i], ql[i] = 1 + i, 1
            continue
        a, b = 1 + rng.below(n_vars - 1), 1 + rng.below(n_vars - 1)
        kind = rng.below(3)
        if kind == 0:                                             # a * b - c = 0 with c a fresh value: reuse a signal through the constant
            c = 1 + rng.below(n_vars - 1)
            A[i], B[i], C[i] = a, b, c
            qm[i], qo[i] = 1, P - 1
            qc[i] = (pool[c] - pool[a] * pool[b]) % P
        elif kind == 1:                                           # l a + r b + o c + k = 0
            c = 1 + rng.below(n_vars - 1)
            l, r_, o = rng.fr(), rng.fr(), rng.fr()
            A[i], B[i], C[i] = a, b, c
            ql[i], qr[i], qo[i] = l, r_, o
            qc[i] = (-(l * pool[a] + r_ * pool[b] + o * pool[c])) % P
        else:                                                     # m a b + l a + k = 0, c unused
            m, l = rng.fr(), rng.fr()
            A[i], B[i] = a, b
            qm[i], ql[i] = m, l
            qc[i] = (-(m * pool[a] * pool[b] + l * pool[a])) % P
    w = fft_root(power)
    k1, k2 = 2, 3
    assert pow(k1, n, P) != 1 and pow(k2, n, P) != 1 and pow(k1 * inv(k2) % P, n, P) != 1
    # copy constraints: the positions of a signal form one cycle; rows >= n_constraints (before the blinding rows) belong to signal 0
    ks = [1, k1, k2]
    sig = [[0] * n for _ in range(3)]
    wpow = [1] * n
    for i in range(1, n):
        wpow[i] = wpow[i - 1] * w % P
    first, last = {}, {}
    for i in range(n):
        for col, m in enumerate((A, B, C)):
            if i >= n - 2:
                sig[col][i] = ks[col] * wpow[i] % P              # identity: the blinding rows are free
                continue
            s = m[i] if i < n_constraints else 0
            if s in last:
                pc, pi = last[s]
                sig[pc][pi] = ks[col] * wpow[i] % P
            else:
                first[s] = (col, i)
            last[s] = (col, i)
    for s, (pc, pi) in last.items():
        fc, fi = first[s]
        sig[pc][pi] = ks[fc] * wpow[fi] % P
    coef = lambda ev: ifft(ev)
    ext = lambda c: fft(list(c) + [0] * (3 * n))
    pols = {"ql": coef(ql), "qr": coef(qr), "qm": coef(qm), "qo": coef(qo), "qc": coef(qc), "s1": coef(sig[0]), "s2": coef(sig[1]), "s3": coef(sig[2])}
    lag = []
    for i in range(max(n_public, 1)):
        e = [0] * n
        e[i] = 1
        c = coef(e)
        lag.append((c, ext(c)))
    c0 = [0] * (8 * n)
    for j, name in enumerate(("ql", "qr", "qo", "qm", "qc", "s1", "s2", "s3")):
        for i in range(n):
            c0[8 * i + j] = pols[name][i]
    # powers of tau: 9 n points (fflonk_prover.c.hpp:289-293), by repeated multiplication in the exponent
    ptau, t = [], 1
    tab = _fixed_base_table()
    for _ in range(9 * n):
        ptau.append(_fixed_base_mul(tab, t))
        t = t * tau % P
    w3 = pow(5, (P - 1) // 3, P)
    zkey = {"power": power, "n": n, "n_vars": n_vars, "n_public": n_public, "n_additions": n_additions, "n_constraints": n_constraints,
            "k1": k1, "k2": k2, "w3": w3, "w4": fft_root(2), "w8": fft_root(3), "wr": pow(w, pow(3, -1, n), P),
            "additions": additions, "A": A, "B": B, "C": C, "pols": pols, "evals": {k: ext(v) for k, v in pols.items()}, "lagrange": lag,
            "ptau": ptau, "c0": c0, "C0": g1_mul(G1, poly_eval(c0, tau))}
    assert pow(w3, 3, P) == 1 and w3 != 1 and pow(zkey["wr"], 3, P) == w
    witness = pool[:n_vars - n_additions]
    witness[0] = 1                                                # (circom's constant signal; the prover overwrites it with 0)
    return zkey, witness, tau


def _fixed_base_table():
    tab, p = [], G1
    for _ in range(64):                                           # 4-bit windows of the generator
        row, q = [None], None
        for _ in range(15):
            q = g1_add(q, p)
            row.append(q)
        tab.append(row)
        for _ in range(4):
            p = g1_add(p, p)
    return tab


def _fixed_base_mul(tab, k):
    acc = None
    for i in range(64):
        d = (k >> (4 * i)) & 15
        if d:
            acc = g1_add(acc, tab[i][d])
    return acc


def write_zkey(path, z):
    """the byte layout FflonkProver::setZkey reads (fflonk_prover.c.hpp:94-330; header zkey_fflonk.cpp:18-49)"""
    n = z["n"]
    q32 = lambda v: int(v).to_bytes(32, "little")
    hdr = struct.pack("<I", 32) + q32(Q_MOD) + struct.pack("<I", 32) + q32(P) + struct.pack("<5I", z["n_vars"], z["n_public"], n, z["n_additions"], z["n_constraints"])
    hdr += fr_mont_bytes([z["k1"], z["k2"], z["w3"], z["w4"], z["w8"], z["wr"]]) + b"\x00" * 128 + g1_bytes(z["C0"])      # X2 ([tau]_2) is the verifier's: not modelled
    adds = b"".join(struct.pack("<II", s1, s2) + fr_mont_bytes([f1, f2]) for s1, s2, f1, f2 in z["additions"])
    u32s = lambda v: struct.pack("<%dI" % len(v), *v)
    secs = [(1, struct.pack("<I", FFLONK_PROTOCOL_ID)), (SEC["header"], hdr), (SEC["additions"], adds),
            (SEC["amap"], u32s(z["A"])), (SEC["bmap"], u32s(z["B"])), (SEC["cmap"], u32s(z["C"]))]
    for name in ("ql", "qr", "qm", "qo", "qc", "s1", "s2", "s3"):
        secs.append((SEC[name], fr_mont_bytes(z["pols"][name]) + fr_mont_bytes(z["evals"][name])))
    secs.append((SEC["lagrange"], b"".join(fr_mont_bytes(c) + fr_mont_bytes(e) for c, e in z["lagrange"])))
    secs.append((SEC["ptau"], b"".join(g1_bytes(p) for p in z["ptau"])))
    secs.append((SEC["c0"], fr_mont_bytes(z["c0"])))
    write_binfile(path, "zkey", 1, secs)


def write_wtns(path, witness):
    """.wtns: section 1 = n8, prime, nVars (wtns_utils.cpp:13-26); section 2 = the values, 32 little-endian bytes each, NOT Montgomery"""
    write_binfile(path, "wtns", 2, [(1, struct.pack("<I", 32) + int(P).to_bytes(32, "little") + struct.pack("<I", len(witness))),
                                    (2, b"".join(int(v).to_bytes(32, "little") for v in witness))])


# ------------------------------------------------------------------------------------------------ the prover
def prove(z, witness, tau, blinding):
    """FflonkProver::prove (fflonk_prover.c.hpp:437-596).  blinding: b_1 .. b_9 (round1 draws them at random, :645-650).
    Commitments are taken through the trapdoor ([p(tau)]_1 = the multi-exponentiation over the key's powers of tau)."""
    n, power = z["n"], z["power"]
    b = [0] + [v % P for v in blinding]
    w = fft_root(power)
    commit = lambda c: g1_mul(G1, poly_eval(c, tau))
    wit = list(witness)
    wit[0] = 0                                                     # :511
    internal = []
    for s1, s2, f1, f2 in z["additions"]:                          # calculateAdditions :598-612
        internal.append((f1 * wit_get(wit, internal, s1, z) + f2 * wit_get(wit, internal, s2, z)) % P)
    get = lambda s: wit_get(wit, internal, s, z)
    proof = {"polynomials": {}, "evaluations": {}}
    # ---- round 1 (:641-686)
    bufs, pol, ev = {}, {}, {}
    for name, (bf1, bf0) in (("A", (b[2], b[1])), ("B", (b[4], b[3])), ("C", (b[6], b[5]))):     # computeWirePolynomial :697-724
        e = [get(s) for s in z[name]] + [0] * (n - z["n_constraints"])
        e[n - 2], e[n - 1] = bf0, bf1                              # buffers[n - 2] = blindingFactors[1], [n - 1] = blindingFactors[0] of (b2, b1): b1 then b2
        bufs[name] = e
        pol[name] = ifft(e)
        ev[name] = fft(pol[name] + [0] * (3 * n))
    t0 = []
    for i in range(4 * n):                                         # computeT0 :727-788
        a_, b_, c_ = ev["A"][i], ev["B"][i], ev["C"][i]
        pi = 0
        for j in range(z["n_public"]):
            pi = (pi - z["lagrange"][j][1][i] * bufs["A"][j]) % P
        E = z["evals"]
        t0.append((a_ * E["ql"][i] + b_ * E["qr"][i] + a_ * b_ % P * E["qm"][i] + c_ * E["qo"][i] + E["qc"][i] + pi) % P)
    pol["T0"] = div_zh(ifft(t0), n)
    assert degree(pol["T0"]) < 2 * n - 2
    pol["C1"] = interleave([pol["A"], pol["B"], pol["C"], pol["T0"]], 4)
    proof["polynomials"]["C1"] = commit(pol["C1"])
    # ---- round 2 (:814-868)
    tr = Transcript()
    tr.add_point(z["C0"])
    for i in range(z["n_public"]):
        tr.add_scalar(bufs["A"][i])
    tr.add_point(proof["polynomials"]["C1"])
    beta = tr.challenge()
    tr.reset(); tr.add_scalar(beta)
    gamma = tr.challenge()
    E = z["evals"]
    num, den = [], []
    wi = 1
    for i in range(n):                                             # computeZ :871-964
        bw = beta * wi % P
        num.append((bufs["A"][i] + bw + gamma) * (bufs["B"][i] + z["k1"] * bw + gamma) % P * (bufs["C"][i] + z["k2"] * bw + gamma) % P)
        den.append((bufs["A"][i] + beta * E["s1"][4 * i] + gamma) * (bufs["B"][i] + beta * E["s2"][4 * i] + gamma) % P * (bufs["C"][i] + beta * E["s3"][4 * i] + gamma) % P)
        wi = wi * w % P
    zbuf, pn, pd = [0] * n, 1, 1
    for i in range(n):                                             # Z[i + 1] = prod_{k <= i} num_k / den_k, Z[0] = the whole product (= 1)
        pn, pd = pn * num[i] % P, pd * den[i] % P
        zbuf[(i + 1) % n] = pn * inv(pd) % P
    if zbuf[0] != 1:
        raise ValueError("Copy constraints does not match")
    pol["Z"] = ifft(zbuf) + [0, 0, 0]                              # fromEvaluations(..., domainSize, 3): three more coefficients for the blinding
    ev["Z"] = fft(pol["Z"][:n] + [0] * (3 * n))
    bf = [b[9], b[8], b[7]]                                        # blindCoefficients :148-157
    for i in range(3):
        pol["Z"][n + i] = (pol["Z"][n + i] + bf[i]) % P
        pol["Z"][i] = (pol["Z"][i] - bf[i]) % P
    assert degree(pol["Z"]) < n + 3
    w2n, w4n = fft_root(power + 1), fft_root(power + 2)
    t1, t1z = [], []
    om = 1
    for i in range(2 * n):                                         # computeT1 :982-1034
        zp = (b[7] * om % P * om + b[8] * om + b[9]) % P
        l1 = z["lagrange"][0][1][2 * i]
        t1.append((ev["Z"][2 * i] - 1) * l1 % P)
        t1z.append(zp * l1 % P)
        om = om * w2n % P
    pol["T1"] = div_zh(ifft(t1), n, 2)
    t1zc = ifft(t1z)
    pol["T1"] = [(x + y) % P for x, y in zip(pol["T1"], t1zc)]
    assert degree(pol["T1"]) < n + 2
    t2, t2z = [], []
    om = 1
    for i in range(4 * n):                                         # computeT2 :1037-1131
        omw = om * w % P
        a_, b_, c_, z_ = ev["A"][i], ev["B"][i], ev["C"][i], ev["Z"][i]
        zw_ = ev["Z"][(4 * n + 4 + i) % (4 * n)]
        zp = (b[7] * om % P * om + b[8] * om + b[9]) % P
        zwp = (b[7] * omw % P * omw + b[8] * omw + b[9]) % P
        bx = beta * om % P
        e1 = (a_ + bx + gamma) * (b_ + bx * z["k1"] + gamma) % P * (c_ + bx * z["k2"] + gamma) % P
        e2 = (a_ + beta * E["s1"][i] + gamma) * (b_ + beta * E["s2"][i] + gamma) % P * (c_ + beta * E["s3"][i] + gamma) % P
        t2.append((e1 * z_ - e2 * zw_) % P)
        t2z.append((e1 * zp - e2 * zwp) % P)
        om = om * w4n % P
    pol["T2"] = div_zh(ifft(t2), n)
    t2zc = ifft(t2z)
    pol["T2"] = [(x + y) % P for x, y in zip(pol["T2"], t2zc)]
    assert degree(pol["T2"]) < 3 * n
    pol["C2"] = interleave([pol["Z"], pol["T1"], pol["T2"]], 3)
    assert degree(pol["C2"]) < 9 * n
    proof["polynomials"]["C2"] = commit(pol["C2"])
    # ---- round 3 (:1153-1241)
    tr.reset(); tr.add_scalar(gamma); tr.add_point(proof["polynomials"]["C2"])
    xi_seed = tr.challenge()
    roots = roots_of(z, xi_seed)
    xi = roots["xi"]
    xiw = xi * w % P
    evs = proof["evaluations"]
    for key, name in (("ql", "ql"), ("qr", "qr"), ("qm", "qm"), ("qo", "qo"), ("qc", "qc"), ("s1", "s1"), ("s2", "s2"), ("s3", "s3")):
        evs[key] = poly_eval(z["pols"][name], xi)
    evs["a"], evs["b"], evs["c"], evs["z"] = (poly_eval(pol[k], xi) for k in ("A", "B", "C", "Z"))
    evs["zw"], evs["t1w"], evs["t2w"] = (poly_eval(pol[k], xiw) for k in ("Z", "T1", "T2"))
    # ---- round 4 (:1244-1289)
    tr.reset(); tr.add_scalar(xi_seed)
    for key in ("ql", "qr", "qm", "qo", "qc", "s1", "s2", "s3", "a", "b", "c", "z", "zw", "t1w", "t2w"):
        tr.add_scalar(evs[key])
    alpha = tr.challenge()
    r0 = lagrange_interpolation(roots["S0h0"], [poly_eval(z["c0"], h) for h in roots["S0h0"]])
    r1 = lagrange_interpolation(roots["S1h1"], [poly_eval(pol["C1"], h) for h in roots["S1h1"]])
    s2 = roots["S2h2"] + roots["S2h3"]
    r2 = lagrange_interpolation(s2, [poly_eval(pol["C2"], h) for h in s2])
    sub = lambda c, r: [(v - (r[i] if i < len(r) else 0)) % P for i, v in enumerate(c)]
    scale = lambda c, k: [v * k % P for v in c]
    F = scale(sub(pol["C2"], r2), alpha * alpha % P)               # computeF :1370-1401
    F = div_by_zerofier(div_by_zerofier(F, 3, xi), 3, xiw)
    f1 = div_by_zerofier(scale(sub(pol["C1"], r1), alpha), 4, xi)
    f0 = div_by_zerofier(sub(z["c0"], r0), 8, xi)
    F = [(F[i] + (f1[i] if i < len(f1) else 0) + (f0[i] if i < len(f0) else 0)) % P for i in range(len(F))]
    assert degree(F) < 9 * n - 6
    proof["polynomials"]["W1"] = commit(F)
    # ---- round 5 (:1417-1459)
    tr.reset(); tr.add_scalar(alpha); tr.add_point(proof["polynomials"]["W1"])
    y = tr.challenge()
    mul = lambda hs: _prod((y - h) % P for h in hs)
    mulL0, mulL1, mulL2 = mul(roots["S0h0"]), mul(roots["S1h1"]), mul(s2)
    preL0, preL1, preL2 = mulL1 * mulL2 % P, alpha * mulL0 % P * mulL2 % P, alpha * alpha % P * mulL0 % P * mulL1 % P
    L = list(pol["C2"])                                            # computeL :1462-1528
    L[0] = (L[0] - poly_eval(r2, y)) % P
    L = scale(L, preL2)
    t = list(pol["C1"]); t[0] = (t[0] - poly_eval(r1, y)) % P
    t = scale(t, preL1)
    u = list(z["c0"]); u[0] = (u[0] - poly_eval(r0, y)) % P
    u = scale(u, preL0)
    zt_y = poly_eval(zerofier(roots["S0h0"] + roots["S1h1"] + s2), y)
    L = [(L[i] + (t[i] if i < len(t) else 0) + (u[i] if i < len(u) else 0) - zt_y * F[i]) % P for i in range(len(L))]
    zts2_y = poly_eval(zerofier(roots["S1h1"] + s2), y)
    L = div_by_zerofier(scale(L, inv(zts2_y)), 1, y)
    assert degree(L) < 9 * n - 1
    proof["polynomials"]["W2"] = commit(L)
    # ---- "inv": the one inversion the verifier is spared (getMontgomeryBatchedInverse :1571-1611)
    to_inv = [mulL1, mulL2, (pow(xi, n, P) - 1) % P]
    to_inv += li_denominators(roots, y, xi, xiw)
    wj = 1
    for _ in range(max(1, z["n_public"])):
        to_inv.append(n * (xi - wj) % P)
        wj = wj * w % P
    evs["inv"] = inv(_prod(to_inv))
    publics = [wit[i] for i in range(1, z["n_public"] + 1)]
    aux = {"beta": beta, "gamma": gamma, "xi_seed": xi_seed, "xi": xi, "alpha": alpha, "y": y, "pol": pol}
    return proof, publics, aux


def wit_get(wit, internal, idx, z):
    """getWitness (:615-629)"""
    diff = z["n_vars"] - z["n_additions"]
    if idx < diff:
        return wit[idx]
    if idx < z["n_vars"]:
        return internal[idx - diff]
    return 0


def _prod(it):
    r = 1
    for v in it:
        r = r * v % P
    return r


def roots_of(z, xi_seed):
    """round3 (:1165-1208): h0 = seed^3 (8 roots of xi), h1 = seed^6 (4), h2 = seed^8 (3), h3 = h2 * omega^(1/3) (3 roots of xi omega); xi = seed^24"""
    w8 = [pow(z["w8"], i, P) for i in range(8)]
    w4 = [pow(z["w4"], i, P) for i in range(4)]
    w3 = [pow(z["w3"], i, P) for i in range(3)]
    h0 = pow(xi_seed, 3, P)
    h1 = h0 * h0 % P
    h2 = h1 * xi_seed % P * xi_seed % P
    h3 = h2 * z["wr"] % P
    return {"S0h0": [h0 * v % P for v in w8], "S1h1": [h1 * v % P for v in w4], "S2h2": [h2 * v % P for v in w3], "S2h3": [h3 * v % P for v in w3],
            "xi": pow(h2, 3, P)}


def li_denominators(roots, y, xi, xiw):
    """computeLiS0 / LiS1 / LiS2 (:1613-1681)"""
    out = []
    h0, h1, h2, h3 = roots["S0h0"], roots["S1h1"], roots["S2h2"], roots["S2h3"]
    d = 8 * pow(h0[0], 6, P) % P
    out += [d * h0[(7 * j) % 8] % P * ((y - h0[j]) % P) % P for j in range(8)]
    d = 4 * h1[0] % P * h1[0] % P
    out += [d * h1[(3 * j) % 4] % P * ((y - h1[j]) % P) % P for j in range(4)]
    d = 3 * h2[0] % P * ((xi - xiw) % P) % P
    out += [d * h2[2 * j % 3] % P * ((y - h2[j]) % P) % P for j in range(3)]
    d = 3 * h3[0] % P * ((xiw - xi) % P) % P
    out += [d * h3[2 * j % 3] % P * ((y - h3[j]) % P) % P for j in range(3)]
    return out


def proof_to_json(proof, publics):
    """SnarkProof::toJson (snark_proof.c.hpp:62-88) and the public signals (fflonk_prover.c.hpp:546-553)"""
    pt = lambda p: ["0", "1", "0"] if p is None else [str(p[0]), str(p[1]), "1"]
    return ({"polynomials": {k: pt(v) for k, v in sorted(proof["polynomials"].items())},
             "evaluations": {k: str(v) for k, v in sorted(proof["evaluations"].items())}, "protocol": "fflonk", "curve": "bn128"}, [str(v) for v in publics])


# ------------------------------------------------------------------------------------------------ the verifier, with the trapdoor for the pairing
def verify(z, tau, proof_json, publics_json):
    """fflonk verification on the JSON proof, written from what the prover's rounds commit to (no verifier is in the reference snapshot): recompute the challenges, the values the three
    combined polynomials must take at the roots (from the claimed evaluations and the circuit identities), interpolate r0, r1, r2 at y and check
        [preL0] C0 + [preL1] C1 + [preL2] C2 - [preL0 r0(y) + preL1 r1(y) + preL2 r2(y)] G - [ZT(y)] W1  ==  [ZTS2(y) (tau - y)] W2
    (the pairing e(lhs, [1]_2) = e([ZTS2(y)] W2, [tau - y]_2) read in G1 through the known tau).  Also checks `inv`."""
    n, power = z["n"], z["power"]
    w = fft_root(power)
    pt = lambda v: None if v[2] == "0" else (int(v[0]), int(v[1]))
    C1, C2, W1, W2 = (pt(proof_json["polynomials"][k]) for k in ("C1", "C2", "W1", "W2"))
    e = {k: int(v) for k, v in proof_json["evaluations"].items()}
    pub = [int(v) for v in publics_json]
    tr = Transcript()
    tr.add_point(z["C0"])
    for v in pub:
        tr.add_scalar(v)
    tr.add_point(C1)
    beta = tr.challenge()
    tr.reset(); tr.add_scalar(beta)
    gamma = tr.challenge()
    tr.reset(); tr.add_scalar(gamma); tr.add_point(C2)
    xi_seed = tr.challenge()
    tr.reset(); tr.add_scalar(xi_seed)
    for key in ("ql", "qr", "qm", "qo", "qc", "s1", "s2", "s3", "a", "b", "c", "z", "zw", "t1w", "t2w"):
        tr.add_scalar(e[key])
    alpha = tr.challenge()
    tr.reset(); tr.add_scalar(alpha); tr.add_point(W1)
    y = tr.challenge()
    roots = roots_of(z, xi_seed)
    xi = roots["xi"]
    xiw = xi * w % P
    zh = (pow(xi, n, P) - 1) % P
    zhi = inv(zh)
    lag = lambda i: pow(w, i, P) * zh % P * inv(n * (xi - pow(w, i, P)) % P) % P          # L_{i+1}(xi)
    pi = 0
    for i, v in enumerate(pub):
        pi = (pi - v * lag(i)) % P
    t0 = (e["ql"] * e["a"] + e["qr"] * e["b"] + e["qm"] * e["a"] % P * e["b"] + e["qo"] * e["c"] + e["qc"] + pi) % P * zhi % P
    t1 = (e["z"] - 1) * lag(0) % P * zhi % P
    t2 = ((e["a"] + beta * xi + gamma) * (e["b"] + beta * xi % P * z["k1"] + gamma) % P * (e["c"] + beta * xi % P * z["k2"] + gamma) % P * e["z"]
          - (e["a"] + beta * e["s1"] + gamma) * (e["b"] + beta * e["s2"] + gamma) % P * (e["c"] + beta * e["s3"] + gamma) % P * e["zw"]) % P * zhi % P
    comb = lambda h, vals: sum(v * pow(h, j, P) for j, v in enumerate(vals)) % P
    y0 = [comb(h, [e["ql"], e["qr"], e["qo"], e["qm"], e["qc"], e["s1"], e["s2"], e["s3"]]) for h in roots["S0h0"]]
    y1 = [comb(h, [e["a"], e["b"], e["c"], t0]) for h in roots["S1h1"]]
    y2 = [comb(h, [e["z"], t1, t2]) for h in roots["S2h2"]] + [comb(h, [e["zw"], e["t1w"], e["t2w"]]) for h in roots["S2h3"]]
    s2 = roots["S2h2"] + roots["S2h3"]
    r0y = poly_eval(lagrange_interpolation(roots["S0h0"], y0), y)
    r1y = poly_eval(lagrange_interpolation(roots["S1h1"], y1), y)
    r2y = poly_eval(lagrange_interpolation(s2, y2), y)
    mul = lambda hs: _prod((y - h) % P for h in hs)
    mulL0, mulL1, mulL2 = mul(roots["S0h0"]), mul(roots["S1h1"]), mul(s2)
    preL0, preL1, preL2 = mulL1 * mulL2 % P, alpha * mulL0 % P * mulL2 % P, alpha * alpha % P * mulL0 % P * mulL1 % P
    zt_y = mulL0 * mulL1 % P * mulL2 % P
    zts2_y = mulL1 * mulL2 % P
    lhs = g1_add(g1_add(g1_mul(z["C0"], preL0), g1_mul(C1, preL1)), g1_mul(C2, preL2))
    lhs = g1_add(lhs, g1_mul(G1, (-(preL0 * r0y + preL1 * r1y + preL2 * r2y)) % P))
    lhs = g1_add(lhs, g1_mul(W1, (-zt_y) % P))
    rhs = g1_mul(W2, zts2_y * (tau - y) % P)
    if lhs != rhs:
        return False
    to_inv = [mulL1, mulL2, zh] + li_denominators(roots, y, xi, xiw)
    wj = 1
    for _ in range(max(1, len(pub))):
        to_inv.append(n * (xi - wj) % P)
        wj = wj * w % P
    return e["inv"] * _prod(to_inv) % P == 1


def random_blinding():
    return [int.from_bytes(os.urandom(31), "little") for _ in range(9)]      # randombytes_buf(v, sizeof(FrElement) - 1) (:649)
