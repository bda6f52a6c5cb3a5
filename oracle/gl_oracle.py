"""ctypes binding of oracle/libgl_oracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never by the product package (xlayer_prover_amd).  See gl_oracle.c header
for provenance and pinning.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgl_oracle.so")
P = 0xFFFFFFFF00000001
u64 = C.c_uint64
_pu64 = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "gl_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    build()
    lib = C.CDLL(_SO)
    sig = {
        "orc_add": (u64, [u64, u64]), "orc_sub": (u64, [u64, u64]), "orc_mul": (u64, [u64, u64]), "orc_mul_fast": (u64, [u64, u64]),
        "orc_inv": (u64, [u64]), "orc_pow": (u64, [u64, u64]), "orc_w": (u64, [C.c_uint]),
        "orc_shift": (u64, []),
        "orc_mul3": (None, [_pu64, _pu64, _pu64]),
        "orc_poseidon_perm": (None, [_pu64, _pu64]),
        "orc_linear_hash": (None, [_pu64, _pu64, u64]),
        "orc_merkletree": (None, [_pu64, _pu64, u64, u64]),
        "orc_merkle_proof": (None, [_pu64, _pu64, _pu64, u64, u64, u64]),
        "orc_merkle_root_from_path": (None, [_pu64, _pu64, u64, _pu64, u64, u64]),
        "orc_find_merkle_index": (u64, [_pu64, u64, _pu64, u64, _pu64]),
        "orc_ntt": (None, [_pu64, _pu64, u64, u64, C.c_int]),
        "orc_dft_naive": (None, [_pu64, _pu64, u64, C.c_int]),
        "orc_extend_pol": (None, [_pu64, _pu64, u64, u64, u64]),
        "orc_extend_pol_team": (None, [_pu64, _pu64, u64, u64, u64]),
        "orc_extend_naive": (None, [_pu64, _pu64, u64, u64]),
        "orc_barycentric_eval": (u64, [_pu64, u64, u64, u64, u64]),
        "orc_fri_fold": (None, [_pu64, _pu64, C.c_uint, C.c_uint, _pu64, u64]),
        "orc_fri_fold_one": (None, [_pu64, _pu64, C.c_uint, C.c_uint, u64, _pu64, u64]),
        "orc_transpose3": (None, [_pu64, _pu64, u64, C.c_uint]),
        "orc_fill": (None, [_pu64, u64, u64, u64]),
        "orc_num_threads": (C.c_int, []),
        "orc_inv3": (None, [_pu64, _pu64]),
        "orc_batch_inverse": (None, [_pu64, _pu64, u64, C.c_uint]),
        "orc_calculate_z": (None, [_pu64, _pu64, _pu64, u64]),
        "orc_calculate_h1h2": (C.c_int, [_pu64, _pu64, _pu64, _pu64, u64, C.c_uint]),
        "orc_geometric": (None, [_pu64, u64, C.c_uint, _pu64, _pu64]),
        "orc_q_split": (None, [_pu64, _pu64, u64, u64, C.c_uint, C.c_uint, u64]),
        "orc_x_div_x_sub_xi": (None, [_pu64, _pu64, _pu64, u64]),
        "orc_transcript_init": (None, [C.c_void_p]),
        "orc_transcript_put": (None, [C.c_void_p, _pu64, u64]),
        "orc_transcript_get1": (u64, [C.c_void_p]),
        "orc_transcript_get_permutations": (None, [C.c_void_p, _pu64, u64, u64]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    return lib


lib = _load()


def _a(x):
    if isinstance(x, (list, tuple)) and x and isinstance(x[0], (int, np.integer)):
        return np.array([int(v) for v in x], dtype=np.uint64)          # exact (a dtype-less conversion of mixed magnitudes goes through float64)
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


def _vec(x):
    """a few field elements -> exact uint64 vector"""
    if isinstance(x, np.ndarray):
        assert x.dtype.kind in "ui"
        return np.ascontiguousarray(x, dtype=np.uint64).ravel()
    if isinstance(x, (int, np.integer)):
        x = [x]
    return np.array([int(v) for v in x], dtype=np.uint64)


def add(a, b): return lib.orc_add(a, b)
def sub(a, b): return lib.orc_sub(a, b)
def mul(a, b): return lib.orc_mul(a, b)
def mul_fast(a, b): return lib.orc_mul_fast(a, b)
def inv(a): return lib.orc_inv(a)
def pow_(a, e): return lib.orc_pow(a, e)
def w(n): return lib.orc_w(n)
def shift(): return lib.orc_shift()


def mul3(a, b):
    o = np.zeros(3, np.uint64)
    lib.orc_mul3(o, _a(a), _a(b))
    return o


def perm(state12):
    o = np.zeros(12, np.uint64)
    lib.orc_poseidon_perm(o, _a(state12))
    return o


def linear_hash(vals):
    v = _a(vals).ravel()
    o = np.zeros(4, np.uint64)
    lib.orc_linear_hash(o, v if v.size else np.zeros(1, np.uint64), v.size)
    return o


def merkletree(src, w_, h):
    src = _a(src).ravel()
    nodes = np.zeros((2 * h - 1) * 4, np.uint64)
    lib.orc_merkletree(nodes, src if src.size else np.zeros(1, np.uint64), w_, h)
    return nodes


def merkle_proof(nodes, src, w_, h, idx):
    nsib = max(h.bit_length() - 1, 0)
    proof = np.zeros(max(w_ + 4 * nsib, 1), np.uint64)
    s = _a(src).ravel()
    lib.orc_merkle_proof(proof, _a(nodes), s if s.size else np.zeros(1, np.uint64), w_, h, idx)
    return proof[: w_ + 4 * nsib]


def merkle_root_from_path(row, sib, idx):
    row = _a(row).ravel()
    sib = _a(sib).ravel()
    o = np.zeros(4, np.uint64)
    lib.orc_merkle_root_from_path(o, row if row.size else np.zeros(1, np.uint64), row.size,
                                  sib if sib.size else np.zeros(1, np.uint64), sib.size // 4, idx)
    return o


def find_merkle_index(row, sib, root):
    row = _a(row).ravel()
    sib = _a(sib).ravel()
    return lib.orc_find_merkle_index(row, row.size, sib, sib.size // 4, _a(root))


def ntt(src, n, ncols, inverse=False):
    src = _a(src).ravel()
    dst = np.zeros(n * ncols, np.uint64)
    if n * ncols:
        lib.orc_ntt(dst, src, n, ncols, 1 if inverse else 0)
    return dst.reshape(n, ncols)


def dft_naive(col, inverse=False):
    col = _a(col).ravel()
    dst = np.zeros(col.size, np.uint64)
    lib.orc_dft_naive(dst, col, col.size, 1 if inverse else 0)
    return dst


def extend_pol(src, n_ext, n, ncols):
    src = _a(src).ravel()
    out = np.zeros(n_ext * ncols, np.uint64)
    if n * ncols:
        lib.orc_extend_pol(out, src, n_ext, n, ncols)
    return out.reshape(n_ext, ncols)


def extend_pol_team(src, n_ext, n, ncols):
    """extend_pol with all host threads on one column at a time (the CPU-baseline form: a narrow trace still uses every core)"""
    src = _a(src).ravel()
    out = np.zeros(n_ext * ncols, np.uint64)
    if n * ncols:
        lib.orc_extend_pol_team(out, src, n_ext, n, ncols)
    return out.reshape(n_ext, ncols)


def barycentric_eval(evals, n, stride, shift_, z):
    """P(z) of the interpolant of evals[i * stride], i < n, on the coset shift * <w(log2 n)>: barycentric formula, no transform (an independent check of extendPol)"""
    return lib.orc_barycentric_eval(_a(evals).ravel(), n, stride, shift_, z)


def extend_naive(col, n_ext):
    col = _a(col).ravel()
    out = np.zeros(n_ext, np.uint64)
    lib.orc_extend_naive(out, col, n_ext, col.size)
    return out


def fri_fold(pol, bits, red, x, shift_inv):
    pol = _a(pol).ravel()
    out = np.zeros(3 << (bits - red), np.uint64)
    lib.orc_fri_fold(out, pol, bits, red, _a(x), shift_inv)
    return out.reshape(-1, 3)


def fri_fold_one(vals, bits, red, g, x, shift_inv):
    o = np.zeros(3, np.uint64)
    lib.orc_fri_fold_one(o, _a(vals).ravel(), bits, red, g, _a(x), shift_inv)
    return o


def transpose3(src, n, tbits):
    dst = np.zeros(3 * n, np.uint64)
    lib.orc_transpose3(dst, _a(src).ravel(), n, tbits)
    return dst.reshape(n, 3)


def fill(n, seed, start=0):
    dst = np.zeros(n, np.uint64)
    if n:
        lib.orc_fill(dst, n, seed, start)
    return dst


def inv3(a):
    o = np.zeros(3, np.uint64)
    lib.orc_inv3(o, _a(a))
    return o


def batch_inverse(src, dim=3):
    src = _a(src).ravel()
    out = np.zeros_like(src)
    if src.size:
        lib.orc_batch_inverse(out, src, src.size // dim, dim)
    return out.reshape(-1, dim) if dim == 3 else out


def calculate_z(num, den):
    num, den = _a(num).ravel(), _a(den).ravel()
    z = np.zeros_like(num)
    if num.size:
        lib.orc_calculate_z(z, num, den, num.size // 3)
    return z.reshape(-1, 3)


def calculate_h1h2(f, t, dim=1):
    """-> (h1, h2) or raises ValueError if some f value is not in t"""
    f, t = _a(f).ravel(), _a(t).ravel()
    n = t.size // dim
    h1, h2 = np.zeros_like(t), np.zeros_like(t)
    if n and lib.orc_calculate_h1h2(h1, h2, f, t, n, dim):
        raise ValueError("calculateH1H2: number not included")
    return (h1.reshape(n, dim), h2.reshape(n, dim)) if dim == 3 else (h1, h2)


def q_split(qq1, n, n_ext, qdeg, qdim, shift_in):
    out = np.zeros(n_ext * qdeg * qdim, np.uint64)
    lib.orc_q_split(out, _a(qq1).ravel(), n, n_ext, qdeg, qdim, shift_in)
    return out


def geometric(n, dim, first, ratio):
    out = np.zeros(max(n * dim, 1), np.uint64)
    lib.orc_geometric(out, n, dim, _vec(first), _vec(ratio))
    return out[: n * dim].reshape(n, dim) if dim == 3 else out[:n]


def x_div_x_sub_xi(x, xi):
    x = _a(x).ravel()
    out = np.zeros(max(3 * x.size, 1), np.uint64)
    if x.size:
        lib.orc_x_div_x_sub_xi(out, x, _a(xi), x.size)
    return out[: 3 * x.size].reshape(-1, 3)


def evmap(LEv, LpEv, n, extend_bits, entries):
    """entries: list of (ndarray section [rows, width], column, dim, prime)"""
    nev = len(entries)
    evals = np.zeros(3 * nev, np.uint64)
    keep = [_a(e[0]) for e in entries]
    ptrs = (C.c_void_p * nev)(*[k.ctypes.data + 8 * e[1] for k, e in zip(keep, entries)])
    strides = _a([k.shape[1] for k in keep])
    dims = np.ascontiguousarray([e[2] for e in entries], dtype=np.uint32)
    primes = np.ascontiguousarray([1 if e[3] else 0 for e in entries], dtype=np.uint32)
    f = lib.orc_evmap
    f.restype = None
    f.argtypes = [_pu64, _pu64, _pu64, u64, C.c_uint, C.c_uint, C.c_void_p, _pu64,
                  np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS"), np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")]
    f(evals, _a(LEv).ravel(), _a(LpEv).ravel(), n, extend_bits, nev, ptrs, strides, dims, primes)
    return evals.reshape(nev, 3)


class OrcStepsParams(C.Structure):
    """orc_steps_params (gl_oracle.c): host pointers + sizes in elements"""
    _fields_ = [("pols", C.c_void_p), ("pols_size", u64),
                ("const_pols", C.c_void_p), ("n_const_pols", u64), ("const_pols_size", u64),
                ("const_pols_2ns", C.c_void_p), ("n_const_pols_2ns", u64), ("const_pols_2ns_size", u64),
                ("challenges", C.c_void_p), ("challenges_size", u64),
                ("public_inputs", C.c_void_p), ("public_inputs_size", u64),
                ("evals", C.c_void_p), ("evals_size", u64),
                ("x_n", C.c_void_p), ("x_n_size", u64),
                ("x_2ns", C.c_void_p), ("x_2ns_size", u64),
                ("zi", C.c_void_p), ("zi_size", u64),
                ("x_div_x_sub_xi", C.c_void_p), ("x_div_x_sub_xi_size", u64),
                ("x_div_x_sub_wxi", C.c_void_p), ("x_div_x_sub_wxi_size", u64),
                ("q_2ns", C.c_void_p), ("q_2ns_size", u64),
                ("f_2ns", C.c_void_p), ("f_2ns_size", u64)]


def expr_run(code, ntmp1, ntmp3, buffers, row0, nrows, n_const_pols=0, n_const_pols_2ns=0):
    """buffers: name -> C-contiguous uint64 ndarray (modified in place for pols / q_2ns / f_2ns)"""
    sp = OrcStepsParams()
    sp.n_const_pols, sp.n_const_pols_2ns = n_const_pols, n_const_pols_2ns
    for name, a in buffers.items():
        assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
        setattr(sp, name, a.ctypes.data)
        setattr(sp, name + "_size", a.size)
    code = _a(code)
    f = lib.orc_expr_run
    f.restype = u64
    f.argtypes = [_pu64, u64, u64, u64, C.c_void_p, u64, u64]
    bad = f(code if code.size else np.zeros(1, np.uint64), code.size, ntmp1, ntmp3, C.byref(sp), row0, nrows)
    if bad:
        raise ValueError(f"expr_run: malformed instruction {bad - 1}")


def num_threads():
    return lib.orc_num_threads()


class Transcript:
    """Mirror of reference Transcript (transcript.hpp:13-37) on the oracle."""

    def __init__(self):
        self._buf = (C.c_uint64 * 26)()
        lib.orc_transcript_init(C.addressof(self._buf))

    def put(self, vals):
        v = _a(vals).ravel()
        if v.size:
            lib.orc_transcript_put(C.addressof(self._buf), v, v.size)

    def get1(self):
        return lib.orc_transcript_get1(C.addressof(self._buf))

    def get_field(self):
        return np.array([self.get1() for _ in range(3)], np.uint64)

    def get_permutations(self, n, nbits):
        res = np.zeros(n, np.uint64)
        lib.orc_transcript_get_permutations(C.addressof(self._buf), res, n, nbits)
        return res
