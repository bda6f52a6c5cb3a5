"""CPU checks of the arithmetic behind the matrix-pipe middle pass of extendPol (xlayer_prover_amd/csrc/ntt_lde_mid_mfma.hip, ntt_mfma_common.hpp):
plain integers, no GPU, no library call.
  1. A DFT16 (any 16 x 16 field matrix) as an int8 matrix product over byte limbs: signed digits of the constants, the xor-0x80 offset (k0), the
     accumulator start digits (bias = a multiple of p, or B0 more), the recombination of eight digit sums -- reproduces the field result.
  2. The round structure of lde_mid_mfma (r1, r2, M, r2', r3': every in-tile factor folded into wave-owned matrices) equals the round structure of the vector kernel
     (ntt_lde_mid_kernel.inc: three inverse rounds, coset scale, three forward rounds with tw2 / twf_12) on a random tile.
The GPU side of the same statement is tests/test_lde_mfma.py (both kernels, same bits)."""
import random

P = 2**64 - 2**32 + 1
B0 = 0x8080808080808080
BOUND = 1 << 21


def inv(a):
    return pow(a, P - 2, P)


def balanced(v):
    d = []
    for _ in range(8):
        dm = v % 256
        if dm >= 128:
            dm -= 256
        d.append(dm)
        v = (v - dm) // 256
    return d if v == 0 else None


def signed_digits(c):
    """ntt_mfma_common.hpp signed_digits: balanced byte digits of c or of c - p"""
    for v in ((c - P, c) if c > P // 2 else (c, c - P)):
        d = balanced(v)
        if d is not None:
            assert sum(x << (8 * m) for m, x in enumerate(d)) % P == c % P and all(-128 <= x <= 127 for x in d)
            return d
    raise AssertionError("no balanced representation")


def bias_digits(off):
    """ntt_mfma_common.hpp bias_digits: digits >= 2^21 whose weighted sum is off (mod p)"""
    base = sum(BOUND << (8 * m) for m in range(8))
    tt = (base - off + P - 1) // P
    R = tt * P + off - base
    out = []
    for _ in range(7):
        out.append(BOUND + (R & 255))
        R >>= 8
    out.append(BOUND + R)
    assert sum(x << (8 * m) for m, x in enumerate(out)) % P == off % P and min(out) >= BOUND and max(out) < (1 << 23) - BOUND
    return out


def recombine(d, a0=0, a1=0):
    """ntt_mfma_common.hpp recombine, with its 32 / 64-bit widths asserted"""
    assert all(0 <= x < (1 << 23) for x in d)
    W0 = (d[1] << 8) + d[0]
    assert W0 < 2**32
    W0 = d[3] * (1 << 24) + d[2] * (1 << 16) + W0 + a0
    W1 = d[7] * (1 << 24) + d[6] * (1 << 16) + (d[5] << 8) + d[4] + a1
    assert W0 < 2**64 and W1 < 2**64
    Z = (W1 >> 32) * 0xFFFFFFFF + W0
    assert Z < 2**64
    hi = (Z >> 32) + (W1 & 0xFFFFFFFF)
    cy, hi = hi >> 32, hi & 0xFFFFFFFF
    r = cy * 0xFFFFFFFF + ((hi << 32) | (Z & 0xFFFFFFFF))
    assert r < 2**64
    return r


def matvec_mfma(M, x, plus_b0, raw_inputs):
    """y = M x through byte limbs.  raw_inputs: the words are plain (the k0 correction applies; M must have row sums 16 e_0);
    otherwise they carry + B0.  plus_b0: the outputs carry + B0."""
    s = [[((xj >> (8 * l)) & 255) - 128 for l in range(8)] for xj in x]
    dig = [[[signed_digits(M[k][j] * pow(256, l, P) % P) for l in range(8)] for j in range(16)] for k in range(16)]
    C = bias_digits(B0 % P if plus_b0 else 0)
    k0 = 16 * (B0 % P) % P
    y = []
    for k in range(16):
        d = []
        for m in range(8):
            a = sum(s[j][l] * dig[k][j][l][m] for j in range(16) for l in range(8))
            assert abs(a) <= BOUND
            d.append(a + C[m])
        y.append(recombine(d, k0 & 0xFFFFFFFF, k0 >> 32) if (raw_inputs and k == 0) else recombine(d))
    return y


W32 = 7277203076849721926
w12 = pow(W32, 2**20, P)
w16 = pow(w12, 256, P)
twf = [pow(w12, i, P) for i in range(4096)]
twi = [pow(inv(w12), i, P) for i in range(4096)]
w16p = [pow(w16, i, P) for i in range(16)]
w16m = [pow(inv(w16), i, P) for i in range(16)]


def test_w16_is_a_power_of_two():
    assert w16 == 4096


def test_dft16_through_byte_limbs():
    rng = random.Random(5)
    F = [[w16p[(j * k) % 16] for j in range(16)] for k in range(16)]
    t = [rng.randrange(P) for _ in range(16)]
    t[0] = 1
    R2 = [[t[k] * w16m[(j * k) % 16] % P for j in range(16)] for k in range(16)]          # row-scaled inverse DFT, factor 1 in row 0
    dense = [[rng.randrange(P) for _ in range(16)] for _ in range(16)]                    # any matrix (inputs + B0, no correction needed)
    edge = [[2**64 - 1] * 16, [0] * 16, [P - 1] * 16, [B0] * 16, [0x7f7f7f7f7f7f7f7f if j & 1 else 2**64 - 1 for j in range(16)], [0x7F] * 16]
    for trial in range(12):
        x = edge[trial] if trial < len(edge) else [rng.getrandbits(64) for _ in range(16)]
        for M in (F, R2):
            ref = [sum(M[k][j] * x[j] for j in range(16)) % P for k in range(16)]
            assert [v % P for v in matvec_mfma(M, x, False, True)] == ref
            assert [(v - B0) % P for v in matvec_mfma(M, x, True, True)] == ref
        ref = [sum(dense[k][j] * (x[j] - B0) for j in range(16)) % P for k in range(16)]       # x carries + B0
        assert [v % P for v in matvec_mfma(dense, x, False, False)] == ref


def nat12(pos):
    return (pos >> 8) | (((pos >> 4) & 15) << 4) | ((pos & 15) << 8)


def vector_kernel_rounds(a, G):
    sk = [pow(G, i, P) for i in range(4096)]
    hk = [pow(G, 256 * j, P) for j in range(16)]
    b = [0] * 4096
    for low in range(256):
        for k in range(16):
            b[low + 256 * k] = sum(a[low + 256 * j] * w16m[(j * k) % 16] for j in range(16)) % P * twi[k * low] % P
    c = [0] * 4096
    for high in range(16):
        for low in range(16):
            base = (high << 8) | low
            for k in range(16):
                c[base + 16 * k] = sum(b[base + 16 * j] * w16m[(j * k) % 16] for j in range(16)) % P * twi[(k * low) << 4] % P
    X = [0] * 4096
    for g in range(256):
        for k in range(16):
            X[16 * g + k] = sum(c[16 * g + j] * w16m[(j * k) % 16] for j in range(16)) % P
    d = [0] * 4096
    for g in range(256):
        for k in range(16):
            d[16 * g + k] = sum(X[16 * g + j] * hk[j] % P * w16p[(j * k) % 16] for j in range(16)) % P
    e = [0] * 4096
    for high in range(16):
        for low in range(16):
            base = (high << 8) | low
            x = [d[base + 16 * j] * (sk[nat12((base + 16 * j) & ~15)] * twf[(j * low) << 4] % P) % P for j in range(16)]
            for k in range(16):
                e[base + 16 * k] = sum(x[j] * w16p[(j * k) % 16] for j in range(16)) % P
    out = [0] * 4096
    for low in range(256):
        x = [e[low + 256 * j] * twf[j * low] % P for j in range(16)]
        for k in range(16):
            out[low + 256 * k] = sum(x[j] * w16p[(j * k) % 16] for j in range(16)) % P
    return out


def matrix_kernel_rounds(a, G):
    """lde_mid_mfma: EVERY in-tile factor folded into a matrix that one wave owns (no vector multiplication left):
       r1  by a = high nibble of low8:  diag_k(w256^-(k a)) F^-1            (the other half of the twiddle, w4096^-(k b), waits)
       r2  by b = low4:                 diag_k(w256^-(k b)) F^-1
       M   by d2:                       diag_k(w4096^(d2 k)) F diag(G^(256 i)) F^-1 diag_j(w4096^-(d2 j))   (takes the waiting half in, puts r3's k0-half out)
       r2' by k0 = low4:                F diag_j(G^(16 j) w256^(j k0))
       r3' by k1 = high nibble of low8: F diag_j(G^j w256^(j k1))"""
    R1 = [[[twi[(k * aa) << 4] * w16m[(j * k) % 16] % P for j in range(16)] for k in range(16)] for aa in range(16)]
    b = [0] * 4096
    for low in range(256):
        aa = low >> 4
        for k in range(16):
            b[low + 256 * k] = sum(R1[aa][k][j] * a[low + 256 * j] for j in range(16)) % P
    R2 = [[[twi[(k * low) << 4] * w16m[(j * k) % 16] % P for j in range(16)] for k in range(16)] for low in range(16)]
    c = [0] * 4096
    for high in range(16):
        for low in range(16):
            base = (high << 8) | low
            for k in range(16):
                c[base + 16 * k] = sum(R2[low][k][j] * b[base + 16 * j] for j in range(16)) % P
    hk = [pow(G, 256 * j, P) for j in range(16)]
    M0 = [[sum(w16p[(i * k) % 16] * hk[i] % P * w16m[(i * j) % 16] for i in range(16)) % P for j in range(16)] for k in range(16)]
    d = [0] * 4096
    for d2 in range(16):
        M = [[twf[d2 * k] * M0[k][j] % P * twi[d2 * j] % P for j in range(16)] for k in range(16)]
        for d1 in range(16):
            base = 256 * d2 + 16 * d1
            for k in range(16):
                d[base + k] = sum(M[k][j] * c[base + j] for j in range(16)) % P
    R2p = [[[w16p[(j * k) % 16] * pow(G, 16 * j, P) % P * twf[(j * low) << 4] % P for j in range(16)] for k in range(16)] for low in range(16)]
    e = [0] * 4096
    for high in range(16):
        for low in range(16):
            base = (high << 8) | low
            for k in range(16):
                e[base + 16 * k] = sum(R2p[low][k][j] * d[base + 16 * j] for j in range(16)) % P
    R3 = [[[w16p[(j * k) % 16] * pow(G, j, P) % P * twf[(j * k1) << 4] % P for j in range(16)] for k in range(16)] for k1 in range(16)]
    out = [0] * 4096
    for low in range(256):
        k1 = low >> 4
        for k in range(16):
            out[low + 256 * k] = sum(R3[k1][k][j] * e[low + 256 * j] for j in range(16)) % P
    return out


def test_round_structure_of_the_matrix_kernel_equals_the_vector_kernel():
    rng = random.Random(4)
    a = [rng.randrange(P) for _ in range(4096)]
    G = rng.randrange(1, P)
    assert vector_kernel_rounds(a, G) == matrix_kernel_rounds(a, G)
