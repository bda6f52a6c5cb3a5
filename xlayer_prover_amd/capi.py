"""ctypes binding of libxlp_gl.so (the C ABI declared in include/xlp_gl.h).

There is deliberately no fallback: if the shared library is missing or no HIP
device can be bound, every call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
P = 0xFFFFFFFF00000001

# name -> (argtypes) ; all return int.  Mirrors include/xlp_gl.h one to one.
_pu = C.c_void_p
_u64 = C.c_uint64
_u32 = C.c_uint32
SIGNATURES = {
    "xlp_init": [C.c_int],
    "xlp_set_stream": [C.c_void_p],
    "xlp_sync": [],
    "xlp_version": [],
    "xlp_ntt": [_pu, _pu, _u64, _u64, C.c_int],
    "xlp_extend": [_pu, _pu, _u64, _u64, _u64, _pu],
    "xlp_merkle": [_pu, _pu, _u64, _u64],
    "xlp_perm": [_pu, _pu],
    "xlp_perm_chain": [_pu, _pu, _pu, _u64],
    "xlp_perm_batch": [_pu, _pu, _u64],
    "xlp_linear_hash": [_pu, _pu, _u64, _u64],
    "xlp_fri_fold": [_pu, _pu, _u32, _u32, _pu, _u64],
    "xlp_transpose3": [_pu, _pu, _u64, _u32],
    "xlp_batch_inverse": [_pu, _pu, _u64, _u32],
    "xlp_calculate_z": [_pu, _pu, _pu, _u64],
    "xlp_calculate_h1h2": [_pu, _pu, _pu, _pu, _u64, _u32],
    "xlp_dev_ntt": [_pu, _pu, _u64, _u64, C.c_int],
    "xlp_dev_extend": [_pu, _pu, _u64, _u64, _u64, _pu],
    "xlp_dev_merkle": [_pu, _pu, _u64, _u64],
    "xlp_dev_perm_batch": [_pu, _pu, _u64],
    "xlp_dev_linear_hash": [_pu, _pu, _u64, _u64],
    "xlp_dev_fri_fold": [_pu, _pu, _u32, _u32, _pu, _u64],
    "xlp_dev_transpose3": [_pu, _pu, _u64, _u32],
    "xlp_dev_fill": [_pu, _u64, _u64, _u64],
    "xlp_dev_gather_rows": [_pu, _pu, _u64, _u64, _u32, _u64],
    "xlp_field_op": [_pu, _pu, _pu, _u64, C.c_int],
    "xlp_dev_batch_inverse": [_pu, _pu, _u64, _u32],
    "xlp_dev_calculate_z": [_pu, _pu, _pu, _u64],
    "xlp_dev_calculate_z_strided": [_pu, _u64, _pu, _u64, _pu, _u64, _u64],
    "xlp_dev_calculate_h1h2_strided": [_pu, _u64, _pu, _u64, _pu, _u64, _pu, _u64, _u64, _u32],
    "xlp_dev_evmap": [_pu, _pu, _pu, _u64, _u32, _u32, _pu],
    "xlp_dev_geometric": [_pu, _u64, _u32, _pu, _pu],
    "xlp_dev_q_split": [_pu, _pu, _u64, _u64, _u32, _u32, _u64],
    "xlp_dev_copy_cols": [_pu, _u64, _pu, _u64, _u64, _u32],
    "xlp_dev_dot_columns": [_pu, _pu, _u64, _u32, _pu, _u64, _u64],
    "xlp_dev_merkle_proofs": [_pu, _pu, _pu, _u64, _u64, _pu, _u64],
    "xlp_dev_x_div_x_sub_xi": [_pu, _pu, _pu, _u64],
    "xlp_dev_calculate_h1h2": [_pu, _pu, _pu, _pu, _u64, _u32],
    "xlp_expr_create": [C.POINTER(C.c_void_p), _pu, _u64, _u64, _u64],
    "xlp_expr_create_ex": [C.POINTER(C.c_void_p), _pu, _u64, _u64, _u64, _u64],
    "xlp_expr_destroy": [_pu],
    "xlp_expr_codegen": [_pu, C.c_char_p],
    "xlp_expr_attach": [_pu, C.c_char_p],
    "xlp_dev_expr_run": [_pu, _pu, _u64, _u64],
    "xlp_profile_enable": [C.c_int],
    "xlp_dev_perm_resident": [_pu, _u64, _u64],
    "xlp_dev_lde_resident": [_pu, _u64, _u64, _u64],
    "xlp_plan_begin": [_u64],
    "xlp_plan_active": [],
    "xlp_plan_report": [_pu, _pu, _pu, _pu, _u32, _pu],
    "xlp_bn128_init": [C.c_char_p],
    "xlp_dev_merkle_bn128": [_pu, _pu, _u64, _u64],
    "xlp_merkle_bn128": [_pu, _pu, _u64, _u64],
    "xlp_dev_poseidon_bn128": [_pu, C.c_uint32, _u64],
    "xlp_poseidon_bn128": [_pu, C.c_uint32, _u64],
    "xlp_dev_merkle_bn128_proofs": [_pu, _pu, _pu, _u64, _u64, _pu, _u64],
    "xlp_dev_fft_bn254": [_pu, _u64, C.c_int],
    "xlp_fft_bn254": [_pu, _u64, C.c_int],
    "xlp_dev_msm_bn254": [_pu, _pu, _pu, _u64],
    "xlp_dev_msm_resident": [_pu, _pu, _u32, _u32],
    "xlp_msm_bn254_window_bits": [_u64],
    "xlp_scratch_release": [_u32],
    "xlp_scratch_lend": [_u32, _pu, C.c_size_t],
    "xlp_dev_fr_gather_mont": [_pu, _pu, _u64, _pu, _u64],
    "xlp_dev_fr_from_mont": [_pu, _pu, _u64],
    "xlp_dev_fr_mul": [_pu, _pu, _pu, _u64],
    "xlp_dev_fr_lincomb": [_pu, _u64, _u32, _pu, _pu, _pu],
    "xlp_dev_fr_interleave": [_pu, _u64, _u32, _pu, _pu],
    "xlp_dev_fr_div_zh": [_pu, _u64, _u64],
    "xlp_dev_fr_batch_inverse": [_pu, _pu, _u64],
    "xlp_dev_fr_prefix_product": [_pu, _pu, _u64],
    "xlp_dev_fr_div_xm_minus": [_pu, _pu, _u64, _u32, _pu],
    "xlp_dev_fr_eval": [_pu, _pu, _u64, _pu, _u32],
    "xlp_dev_fflonk_t0": [_pu] * 11 + [_u32, _u64],
    "xlp_dev_fflonk_z_numden": [_pu] * 8 + [_u64, _pu, _pu, _u32],
    "xlp_dev_fflonk_z_rotate": [_pu, _pu, _u64],
    "xlp_dev_fflonk_t1": [_pu, _pu, _pu, _pu, _u64, _pu, _pu, _u32],
    "xlp_dev_fflonk_t2": [_pu] * 9 + [_u64, _pu, _pu, _u32],
    "xlp_msm_bn254": [_pu, _pu, _pu, _u64],
    "xlp_profile_read": [C.POINTER(C.c_double), C.POINTER(C.c_uint64)],
    "xlp_dev_extend_cols": [_pu, _pu, _u64, _u64, _u64, _u64, _u64],
    "xlp_dev_linear_hash_absorb": [_pu, _pu, _u64, _u64, _u64, _u64, C.c_int, C.c_int],
    "xlp_dev_merkle_levels": [_pu, _u64],
    "xlp_dev_commit": [_pu, _pu, _pu, _u64, _u64, _u64],
    "xlp_dev_commit_last_times": [C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "xlp_dev_evmap_cm": [_pu, _pu, _pu, _u64, C.c_uint32, _pu],
    "xlp_dev_dot_columns_cm": [_pu, _u64, _pu, _u64, C.c_uint32, _pu, _u64, _u64],
    "xlp_dev_rows_to_cm": [_pu, _u64, _u64, _pu, _u64, _u64, _u64, _u64, C.c_uint32],
    "xlp_dev_cm_to_rows": [_pu, _u64, _u64, _pu, _u64, _u64, _u64, _u64, C.c_uint32],
    "xlp_dev_copy_view": [_pu, _u64, _u64, _pu, _u64, _u64, _u64, C.c_uint32],
    "xlp_dev_extend_cm": [_pu, _u64, _u64, _pu, _u64, _u64, _u64, _u64],
    "xlp_dev_copy_3d": [_pu, _u64, _u64, _pu, _u64, _u64, _u64, _u64, _u64],
    "xlp_dev_linear_hash_absorb_cm": [_pu, _pu, _u64, _u64, _u64, _u64, _u64, C.c_uint32, C.c_int, C.c_int],
    "xlp_dev_merkle_cm": [_pu, _pu, _u64, _u64, _u64, _u64, C.c_uint32],
    "xlp_dev_merkle_proofs_cm": [_pu, _pu, _u64, _u64, _pu, _u64, _u64, C.c_uint32, _pu, _u64],
    "xlp_dev_commit_cm": [_pu, _u64, _pu, _pu, _u64, _u64, _u64, _u64],
    "xlp_dev_commit_from_host_cm": [_pu, _u64, _pu, _u64, _pu, _pu, _pu, _u64, C.c_int, _u64, _u64, _u64, C.POINTER(C.c_double)],
    "xlp_dev_commit_from_host": [_pu, _pu, _pu, _pu, _u64, _u64, _u64, C.POINTER(C.c_double)],
    "xlp_cache_enable": [C.c_size_t],
    "xlp_cache_invalidate": [_pu, C.c_size_t],
    "xlp_host_alloc": [C.POINTER(C.c_void_p), C.c_size_t],
    "xlp_host_free": [_pu],
    "xlp_host_register": [_pu, C.c_size_t],
    "xlp_host_unregister": [_pu],
    "xlp_dev_alloc": [C.POINTER(C.c_void_p), C.c_size_t],
    "xlp_dev_mem_info": [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
    "xlp_dev_free": [_pu],
    "xlp_dev_memset": [_pu, C.c_int, C.c_size_t],
    "xlp_copy_h2d": [_pu, _pu, C.c_size_t],
    "xlp_copy_d2h": [_pu, _pu, C.c_size_t],
    "xlp_dev_scale3_strided": [_pu, _u64, _u64, _pu],
    "xlp_dev_upload_begin": [],
    "xlp_dev_upload_cols_async": [_pu, _pu, _u64, _u64, _u64, _u64],
    "xlp_dev_upload_fence": [],
    "xlp_comm_init": [C.c_int, C.c_int, C.c_char_p, C.c_char_p],
    "xlp_comm_rank": [],
    "xlp_comm_world": [],
    "xlp_comm_exchange": [_pu, _pu, _pu, _pu, _u64],
    "xlp_comm_ranks_seen": [],
    "xlp_stream_select": [C.c_int],
    "xlp_stream_wait": [C.c_int, C.c_int],
    "xlp_stream_mark": [C.c_int, C.c_int],
    "xlp_stream_wait_mark": [C.c_int, C.c_int],
    "xlp_comm_all_gather": [_pu, _pu, _u64],
    "xlp_comm_sendrecv": [_pu, C.c_int, _pu, C.c_int, _u64],
    "xlp_comm_all_gather_host": [_pu, _pu, _u64],
    "xlp_comm_barrier": [],
    "xlp_comm_stats": [C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int],
}
VOID_SIGNATURES = {"xlp_shutdown": [], "xlp_comm_shutdown": [], "xlp_comm_abort": [C.c_char_p]}
SIZE_SIGNATURES = {"xlp_cache_bytes": [], "xlp_merkle_bn128_num_nodes": [C.c_uint64]}
STR_SIGNATURES = {"xlp_last_error": []}


class XlpError(RuntimeError):
    pass


def lib_path():
    # XLP_LIB selects an alternative build (ablation studies, tools/ablate.sh); default = the in-tree product
    return os.environ.get("XLP_LIB") or os.path.join(_HERE, "libxlp_gl.so")


def load():
    """dlopen libxlp_gl.so and bind every symbol of include/xlp_gl.h (no GPU needed)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: torch ships its own libamdhip64.so.7; if it is
    # importable, load it FIRST so that our NEEDED libamdhip64.so.7 resolves (by
    # SONAME) to the copy torch uses.  Two runtimes in one process cannot both
    # bind the GPU ("No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise XlpError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    lib = C.CDLL(path)
    for name, args in SIGNATURES.items():
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = C.c_int
    for name, args in VOID_SIGNATURES.items():
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = None
    for name, args in SIZE_SIGNATURES.items():
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = C.c_size_t
    lib.xlp_last_error.argtypes = []
    lib.xlp_last_error.restype = C.c_char_p
    _LIB = lib
    return lib


def _ck(rc, what):
    if rc != 0:
        raise XlpError(f"{what} failed: {load().xlp_last_error().decode()}")


def init(device=0, bind_torch_stream=True):
    """Bind the library to HIP device `device`.  When torch is importable and sees a GPU, the library's work is put on torch's
    current stream (xlp_set_stream), so that torch-side readers of the same buffers (to_np, from_np, .clone(), the caching
    allocator's reuse of a freed tensor) are ordered after the kernels by stream order -- without that the library runs on its
    own non-blocking stream and every caller has to remember xlp_sync()."""
    _ck(load().xlp_init(device), "xlp_init")
    if bind_torch_stream:
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.set_device(device)
                use_torch_stream()
        except ImportError:
            pass


def sync():
    _ck(load().xlp_sync(), "xlp_sync")


def cache_enable(max_bytes):
    """device copies of host buffers for the host-pointer API (xlp_cache_enable); 0 turns it off"""
    _ck(load().xlp_cache_enable(max_bytes), "xlp_cache_enable")


def cache_invalidate(arr=None):
    if arr is None:
        _ck(load().xlp_cache_invalidate(None, 0), "xlp_cache_invalidate")
    else:
        _ck(load().xlp_cache_invalidate(arr.ctypes.data_as(C.c_void_p), arr.nbytes), "xlp_cache_invalidate")


def cache_bytes():
    return load().xlp_cache_bytes()


def shutdown():
    """release the context (tables, scratch slots, stream); a later init() binds again"""
    load().xlp_shutdown()


def profile_enable(on=True):
    _ck(load().xlp_profile_enable(1 if on else 0), "xlp_profile_enable")


def profile_read():
    """-> ([ms_top_dif, ms_mid, ms_top_dit] summed, number of recorded LDE column chunks)"""
    ms = (C.c_double * 3)()
    cnt = C.c_uint64(0)
    _ck(load().xlp_profile_read(ms, C.byref(cnt)), "xlp_profile_read")
    return [ms[0], ms[1], ms[2]], cnt.value


def use_torch_stream():
    """Run xlp_dev_* work on torch's current HIP stream (so torch events bracket it)."""
    import torch
    _ck(load().xlp_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)), "xlp_set_stream")


# ----------------------------------------------------------------- host layer
def _np(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def _u64vec(x):
    """a few field elements (Python ints, a list, an ndarray) -> exact uint64 vector.  NOT via np.atleast_1d / np.asarray
    without a dtype: a list mixing values below and above 2^63 becomes float64 there and loses its low bits."""
    if isinstance(x, np.ndarray):
        assert x.dtype.kind in "ui", "field elements as floating point"
        return np.ascontiguousarray(x, dtype=np.uint64).ravel()
    if isinstance(x, (int, np.integer)):
        x = [x]
    return np.array([int(v) for v in x], dtype=np.uint64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def ntt(src, n, ncols, inverse=False, inplace=False):
    src = _np(src).reshape(-1)
    assert src.size == n * ncols
    dst = src if inplace else np.empty_like(src)
    _ck(load().xlp_ntt(_p(dst), _p(src), n, ncols, 1 if inverse else 0), "xlp_ntt")
    return dst.reshape(n, ncols)


def extend(src, n_ext, n, ncols):
    src = _np(src).reshape(-1)
    assert src.size == n * ncols
    out = np.empty(n_ext * ncols, np.uint64)
    _ck(load().xlp_extend(_p(out), _p(src), n_ext, n, ncols, None), "xlp_extend")
    return out.reshape(n_ext, ncols)


def extend_into(out, src, n_ext, n, ncols):
    """xlp_extend into an existing host array (its address range is what the device-copy cache keys on)"""
    src = _np(src).reshape(-1)
    assert src.size == n * ncols and out.dtype == np.uint64 and out.size == n_ext * ncols and out.flags["C_CONTIGUOUS"]
    _ck(load().xlp_extend(_p(out), _p(src), n_ext, n, ncols, None), "xlp_extend")
    return out.reshape(n_ext, ncols)


def merkle(src, w, h):
    src = _np(src).reshape(-1)
    assert src.size == w * h
    nodes = np.empty((2 * h - 1) * 4, np.uint64)
    _ck(load().xlp_merkle(_p(nodes), _p(src) if src.size else None, w, h), "xlp_merkle")
    return nodes


def merkle_bn128(src, w, h):
    """MerkleTreeBN128::merkelize -> (num_nodes, 4) uint64: Fr elements as Montgomery limbs (the reference's RawFr::Element bytes)"""
    src = _np(src).reshape(-1)
    assert src.size == w * h
    nn = load().xlp_merkle_bn128_num_nodes(h)
    nodes = np.zeros((nn, 4), np.uint64)
    _ck(load().xlp_merkle_bn128(_p(nodes), _p(src) if src.size else None, w, h), "xlp_merkle_bn128")
    return nodes


def poseidon_bn128(states, t):
    """Poseidon_opt::hash on (count, t, 4) uint64 Montgomery-limb states -> permuted states"""
    st = _np(states).reshape(-1, t, 4).copy()
    _ck(load().xlp_poseidon_bn128(_p(st), t, st.shape[0]), "xlp_poseidon_bn128")
    return st


def fft_bn254(data, inverse=False):
    """FFT<RawFr>::fft / ifft on (n, 4) uint64 Montgomery-limb elements -> transformed copy"""
    a = _np(data).reshape(-1, 4).copy()
    _ck(load().xlp_fft_bn254(_p(a), a.shape[0], 1 if inverse else 0), "xlp_fft_bn254")
    return a


def msm_bn254(bases, scalars):
    """sum scalars[i] * bases[i] on BN254 G1: bases (n, 8) uint64 (x, y Montgomery limbs), scalars (n, 4) uint64 raw -> (8,) affine sum"""
    b = _np(bases).reshape(-1, 8)
    s = _np(scalars).reshape(-1, 4)
    assert b.shape[0] == s.shape[0]
    out = np.zeros(8, np.uint64)
    _ck(load().xlp_msm_bn254(_p(out), _p(b) if b.size else None, _p(s) if s.size else None, b.shape[0]), "xlp_msm_bn254")
    return out


def dev_msm_bn254(out, bases, scalars, n):
    _ck(load().xlp_dev_msm_bn254(_dp(out), _dp(bases), _dp(scalars), n), "xlp_dev_msm_bn254")


# ---- vectors of BN254 Fr elements (the fflonk prover's primitives): device tensors of (n, 4) int64 Montgomery limbs; scalars as numpy (k, 4) uint64
def scratch_lend(slot, tensor=None):
    """let a scratch slot of the library use a caller-owned device tensor (None: take it back)"""
    if tensor is None:
        _ck(load().xlp_scratch_lend(slot, None, 0), "xlp_scratch_lend")
    else:
        _ck(load().xlp_scratch_lend(slot, _dp(tensor), tensor.numel() * tensor.element_size()), "xlp_scratch_lend")


def scratch_release(mask=0xFF):
    _ck(load().xlp_scratch_release(mask), "xlp_scratch_release")


def dev_fr_mul(out, a, b, n):
    _ck(load().xlp_dev_fr_mul(_dp(out), _dp(a), _dp(b), n), "xlp_dev_fr_mul")


def dev_fr_from_mont(out, a, n):
    _ck(load().xlp_dev_fr_from_mont(_dp(out), _dp(a), n), "xlp_dev_fr_from_mont")


def dev_fr_batch_inverse(out, a, n):
    _ck(load().xlp_dev_fr_batch_inverse(_dp(out), _dp(a), n), "xlp_dev_fr_batch_inverse")


def dev_fr_prefix_product(out, a, n):
    _ck(load().xlp_dev_fr_prefix_product(_dp(out), _dp(a), n), "xlp_dev_fr_prefix_product")


def dev_fr_div_zh(pol, length, n):
    _ck(load().xlp_dev_fr_div_zh(_dp(pol), length, n), "xlp_dev_fr_div_zh")


def dev_fr_div_xm_minus(out, a, length, m, beta):
    b = np.ascontiguousarray(beta, dtype=np.uint64).reshape(4)
    _ck(load().xlp_dev_fr_div_xm_minus(_dp(out), _dp(a), length, m, _p(b)), "xlp_dev_fr_div_xm_minus")


def dev_fr_eval(pol, length, points):
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros_like(pts)
    _ck(load().xlp_dev_fr_eval(_p(out), _dp(pol), length, _p(pts), pts.shape[0]), "xlp_dev_fr_eval")
    return out


def dev_fr_lincomb(out, length, terms):
    """terms: [(device tensor, offset in elements, len, coefficient (4,) uint64)]"""
    k = len(terms)
    ptrs = (C.c_void_p * k)(*[_dpo(t, 4 * off) for t, off, _, _ in terms])
    lens = np.array([l for _, _, l, _ in terms], dtype=np.uint64)
    coefs = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.uint64).reshape(4) for _, _, _, c in terms]))
    _ck(load().xlp_dev_fr_lincomb(_dp(out), length, k, C.cast(ptrs, C.c_void_p), _p(lens), _p(coefs)), "xlp_dev_fr_lincomb")


def dev_fr_interleave(out, out_len, pols):
    """pols: [(device tensor, len)]"""
    k = len(pols)
    ptrs = (C.c_void_p * k)(*[_dp(t) for t, _ in pols])
    lens = np.array([l for _, l in pols], dtype=np.uint64)
    _ck(load().xlp_dev_fr_interleave(_dp(out), out_len, k, C.cast(ptrs, C.c_void_p), _p(lens)), "xlp_dev_fr_interleave")


def dev_fft_bn254(data, n, inverse=False):
    _ck(load().xlp_dev_fft_bn254(_dp(data), n, 1 if inverse else 0), "xlp_dev_fft_bn254")


def perm(state):
    s = _np(state).reshape(-1)
    assert s.size == 12
    out = np.empty(12, np.uint64)
    _ck(load().xlp_perm(_p(out), _p(s)), "xlp_perm")
    return out


def perm_chain(state4, elems):
    """xlp_perm_chain: the full blocks of a long Transcript::put in one call -> the last permutation's 12 words"""
    st, e = _np(state4).reshape(-1), _np(elems).reshape(-1)
    assert st.size == 4 and e.size % 8 == 0 and e.size
    out = np.empty(12, np.uint64)
    _ck(load().xlp_perm_chain(_p(out), _p(st), _p(e), e.size // 8), "xlp_perm_chain")
    return out


def perm_batch(states):
    s = _np(states).reshape(-1, 12)
    out = np.empty_like(s)
    _ck(load().xlp_perm_batch(_p(out), _p(s), s.shape[0]), "xlp_perm_batch")
    return out


def linear_hash(src, w, h):
    src = _np(src).reshape(-1)
    out = np.empty(4 * h, np.uint64)
    _ck(load().xlp_linear_hash(_p(out), _p(src) if src.size else None, w, h), "xlp_linear_hash")
    return out.reshape(h, 4)


def fri_fold(pol, bits, red, x, shift_inv):
    pol = _np(pol).reshape(-1)
    assert pol.size == 3 << bits
    x = _np(x)
    out = np.empty(3 << (bits - red), np.uint64)
    _ck(load().xlp_fri_fold(_p(out), _p(pol), bits, red, _p(x), int(shift_inv)), "xlp_fri_fold")
    return out.reshape(-1, 3)


def transpose3(src, n, tbits):
    src = _np(src).reshape(-1)
    out = np.empty(3 * n, np.uint64)
    _ck(load().xlp_transpose3(_p(out), _p(src), n, tbits), "xlp_transpose3")
    return out.reshape(n, 3)


def field_op(a, b, op):
    a = _np(a).reshape(-1)
    b = _np(b).reshape(-1)
    out = np.empty_like(a)
    _ck(load().xlp_field_op(_p(out), _p(a), _p(b), a.size, op), "xlp_field_op")
    return out


# --------------------------------------------------------------- device layer
# torch int64 CUDA tensors carry the uint64 words (torch is only the allocator).
def _dp(t):
    return C.c_void_p(t.data_ptr())


def dev_fill(t, seed, start=0):
    _ck(load().xlp_dev_fill(_dp(t), t.numel(), seed, start), "xlp_dev_fill")


def dev_gather_rows(dst, src, src_stride, n, dim, seed):
    _ck(load().xlp_dev_gather_rows(_dp(dst), _dp(src), src_stride, n, dim, seed), "xlp_dev_gather_rows")


def dev_ntt(dst, src, n, ncols, inverse=False):
    _ck(load().xlp_dev_ntt(_dp(dst), _dp(src), n, ncols, 1 if inverse else 0), "xlp_dev_ntt")


def dev_extend(out, src, n_ext, n, ncols, scratch=None):
    _ck(load().xlp_dev_extend(_dp(out), _dp(src), n_ext, n, ncols, _dp(scratch) if scratch is not None else None), "xlp_dev_extend")


def dev_merkle(nodes, src, w, h):
    _ck(load().xlp_dev_merkle(_dp(nodes), _dp(src), w, h), "xlp_dev_merkle")


def merkle_bn128_num_nodes(h):
    return int(load().xlp_merkle_bn128_num_nodes(h))


def dev_merkle_bn128(nodes, src, w, h):
    """MerkleTreeBN128::merkelize on device tensors: nodes = merkle_bn128_num_nodes(h) x 4 words (Fr, Montgomery limbs)"""
    _ck(load().xlp_dev_merkle_bn128(_dp(nodes), _dp(src), w, h), "xlp_dev_merkle_bn128")


def dev_merkle_bn128_proofs(src, nodes, w, h, idxs):
    """-> ndarray [nq][w + 64 ceil(log16 h)]: row, then per level the 16-node group (Montgomery limbs), for every index"""
    ix = np.ascontiguousarray(idxs, dtype=np.uint64)
    levels, c = 0, 1
    while c < h:
        c *= 16; levels += 1
    out = np.zeros((ix.size, w + 64 * levels), np.uint64)
    _ck(load().xlp_dev_merkle_bn128_proofs(out.ctypes.data_as(C.c_void_p), _dp(src), _dp(nodes), w, h, ix.ctypes.data_as(C.c_void_p), ix.size), "xlp_dev_merkle_bn128_proofs")
    return out


def dev_extend_cols(out, src, n_ext, n, ld, col0, ncols):
    _ck(load().xlp_dev_extend_cols(_dp(out), _dp(src), n_ext, n, ld, col0, ncols), "xlp_dev_extend_cols")


def dev_linear_hash_absorb(state, src, ld, col0, ncols, h, first, last):
    _ck(load().xlp_dev_linear_hash_absorb(_dp(state), _dp(src), ld, col0, ncols, h, 1 if first else 0, 1 if last else 0), "xlp_dev_linear_hash_absorb")


def dev_merkle_levels(nodes, h):
    _ck(load().xlp_dev_merkle_levels(_dp(nodes), h), "xlp_dev_merkle_levels")


def dev_commit(sec_2ns, nodes, sec_n, n_ext, n, ncols):
    """extendPol + merkelize of one section in 64-column bands (xlp_dev_commit)"""
    _ck(load().xlp_dev_commit(_dp(sec_2ns), _dp(nodes), _dp(sec_n), n_ext, n, ncols), "xlp_dev_commit")


def dev_commit_from_host(sec_n, sec_2ns, nodes, host, n_ext, n, ncols, want_upload_ms=False):
    """host: HostBuffer (pinned) or a C-contiguous uint64 ndarray holding the n x ncols base-domain rows"""
    ptr = host.ptr if isinstance(host, HostBuffer) else host.ctypes.data_as(C.c_void_p)
    ms = C.c_double(0.0)
    _ck(load().xlp_dev_commit_from_host(_dp(sec_n), _dp(sec_2ns), _dp(nodes), ptr, n_ext, n, ncols, C.byref(ms) if want_upload_ms else None), "xlp_dev_commit_from_host")
    return ms.value


# ---- column-major sections (include/xlp_gl.h): tensors are flat device int64; `off` = word offset of the first column inside the tensor
def _dpo(t, off=0):
    return C.c_void_p(t.data_ptr() + 8 * off)


def dev_rows_to_cm(dst, dst_cstride, src, ld, col0, ncols, rows, ext_bits=0, dst_off=0, coset_stride=0):
    _ck(load().xlp_dev_rows_to_cm(_dpo(dst, dst_off), dst_cstride, coset_stride, _dp(src), ld, col0, ncols, rows, ext_bits), "xlp_dev_rows_to_cm")


def dev_cm_to_rows(dst, ld, col0, src, src_cstride, ncols, rows, ext_bits=0, src_off=0, coset_stride=0):
    _ck(load().xlp_dev_cm_to_rows(_dp(dst), ld, col0, _dpo(src, src_off), src_cstride, coset_stride, ncols, rows, ext_bits), "xlp_dev_cm_to_rows")


def dev_copy_view(dst, drs, dcs, src, srs, scs, n, dim, dst_off=0, src_off=0):
    _ck(load().xlp_dev_copy_view(_dpo(dst, dst_off), drs, dcs, _dpo(src, src_off), srs, scs, n, dim), "xlp_dev_copy_view")


def dev_copy_3d(dst, ds0, ds1, src, ss0, ss1, n0, n1, n, dst_off=0, src_off=0):
    """dst[a * ds0 + b * ds1 + i] = src[a * ss0 + b * ss1 + i] (a < n0, b < n1, i < n): runs of n contiguous words regrouped"""
    _ck(load().xlp_dev_copy_3d(_dpo(dst, dst_off), ds0, ds1, _dpo(src, src_off), ss0, ss1, n0, n1, n), "xlp_dev_copy_3d")


def dev_extend_cm(out, out_cstride, src, in_cstride, n_ext, n, ncols, out_off=0, src_off=0, out_coset_stride=0):
    _ck(load().xlp_dev_extend_cm(_dpo(out, out_off), out_cstride, out_coset_stride, _dpo(src, src_off), in_cstride, n_ext, n, ncols), "xlp_dev_extend_cm")


def dev_linear_hash_absorb_cm(state, src, cstride, col0, ncols, h, ext_bits, first, last, coset_stride=0):
    _ck(load().xlp_dev_linear_hash_absorb_cm(_dp(state), _dp(src), cstride, coset_stride, col0, ncols, h, ext_bits, 1 if first else 0, 1 if last else 0), "xlp_dev_linear_hash_absorb_cm")


def dev_merkle_cm(nodes, src, cstride, w, h, ext_bits, coset_stride=0):
    _ck(load().xlp_dev_merkle_cm(_dp(nodes), _dp(src), cstride, coset_stride, w, h, ext_bits), "xlp_dev_merkle_cm")


def dev_merkle_proofs_cm(src, cstride, nodes, w, h, ext_bits, idxs, coset_stride=0):
    ix = np.ascontiguousarray(idxs, dtype=np.uint64)
    levels = max(h.bit_length() - 1, 0)
    out = np.zeros((ix.size, w + 4 * levels), np.uint64)
    _ck(load().xlp_dev_merkle_proofs_cm(out.ctypes.data_as(C.c_void_p), _dp(src), cstride, coset_stride, _dp(nodes), w, h, ext_bits, ix.ctypes.data_as(C.c_void_p), ix.size),
        "xlp_dev_merkle_proofs_cm")
    return out


def dev_commit_cm(sec_2ns, cs2, nodes, sec_n, csn, n_ext, n, ncols):
    _ck(load().xlp_dev_commit_cm(_dp(sec_2ns), cs2, _dp(nodes), _dp(sec_n), csn, n_ext, n, ncols), "xlp_dev_commit_cm")


def dev_commit_from_host_cm(sec_n, csn, sec_2ns, cs2, nodes, host, stage, n_ext, n, ncols, want_upload_ms=False, clear_stage=False):
    ptr = host.ptr if isinstance(host, HostBuffer) else host.ctypes.data_as(C.c_void_p)
    ms = C.c_double(0.0)
    _ck(load().xlp_dev_commit_from_host_cm(_dp(sec_n), csn, _dp(sec_2ns), cs2, _dp(nodes), ptr, _dp(stage), stage.numel(), 1 if clear_stage else 0, n_ext, n, ncols, C.byref(ms) if want_upload_ms else None),
        "xlp_dev_commit_from_host_cm")
    return ms.value


class HostBuffer:
    """page-locked host memory (xlp_host_alloc); .array is a uint64 ndarray view of it"""

    def __init__(self, words):
        self.ptr = C.c_void_p(None)
        _ck(load().xlp_host_alloc(C.byref(self.ptr), 8 * max(words, 1)), "xlp_host_alloc")
        self.words = words
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint64)), shape=(max(words, 1),))[:words]

    def close(self):
        if self.ptr:
            self.array = None
            load().xlp_host_free(self.ptr)
            self.ptr = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def copy_d2h(host, dev_tensor):
    """device tensor -> HostBuffer / ndarray (synchronous)"""
    ptr = host.ptr if isinstance(host, HostBuffer) else host.ctypes.data_as(C.c_void_p)
    _ck(load().xlp_copy_d2h(ptr, _dp(dev_tensor), 8 * dev_tensor.numel()), "xlp_copy_d2h")


def dev_perm_batch(out, src, count):
    _ck(load().xlp_dev_perm_batch(_dp(out), _dp(src), count), "xlp_dev_perm_batch")


def dev_linear_hash(out, src, w, h):
    _ck(load().xlp_dev_linear_hash(_dp(out), _dp(src), w, h), "xlp_dev_linear_hash")


def dev_fri_fold(pol2, pol, bits, red, x, shift_inv):
    x = _u64vec(x)
    _ck(load().xlp_dev_fri_fold(_dp(pol2), _dp(pol), bits, red, _p(x), int(shift_inv)), "xlp_dev_fri_fold")


def dev_transpose3(dst, src, n, tbits):
    _ck(load().xlp_dev_transpose3(_dp(dst), _dp(src), n, tbits), "xlp_dev_transpose3")


def dev_batch_inverse(res, src, n, dim=3):
    _ck(load().xlp_dev_batch_inverse(_dp(res), _dp(src), n, dim), "xlp_dev_batch_inverse")


def dev_calculate_z(z, num, den, n):
    _ck(load().xlp_dev_calculate_z(_dp(z), _dp(num), _dp(den), n), "xlp_dev_calculate_z")


def dev_calculate_z_strided(z, zs, num, ns, den, ds, n):
    """z / num / den: device tensors whose element i sits at [i * stride, i * stride + 3) (views into row-major sections)"""
    _ck(load().xlp_dev_calculate_z_strided(_dp(z), zs, _dp(num), ns, _dp(den), ds, n), "xlp_dev_calculate_z_strided")


def dev_calculate_h1h2_strided(h1, h1s, h2, h2s, f, fs, t, ts, n, dim=1):
    _ck(load().xlp_dev_calculate_h1h2_strided(_dp(h1), h1s, _dp(h2), h2s, _dp(f), fs, _dp(t), ts, n, dim), "xlp_dev_calculate_h1h2_strided")


def dev_calculate_h1h2(h1, h2, f, t, n, dim=1):
    _ck(load().xlp_dev_calculate_h1h2(_dp(h1), _dp(h2), _dp(f), _dp(t), n, dim), "xlp_dev_calculate_h1h2")


def dev_merkle_proofs(src, nodes, w, h, idxs):
    """-> ndarray [nq][w + 4 log2 h]: row then sibling nodes bottom-up, for every index"""
    ix = np.ascontiguousarray(idxs, dtype=np.uint64)
    levels = max(h.bit_length() - 1, 0)
    out = np.zeros((ix.size, w + 4 * levels), np.uint64)
    _ck(load().xlp_dev_merkle_proofs(out.ctypes.data_as(C.c_void_p), _dp(src), _dp(nodes), w, h, ix.ctypes.data_as(C.c_void_p), ix.size), "xlp_dev_merkle_proofs")
    return out


def dev_copy_cols(dst, dst_stride, src, src_stride, n, dim):
    """dst[i * dst_stride + c] = src[i * src_stride + c], i < n, c < dim (Polinomial::copy between strided views: the
    transposeH1H2Columns / transposeZColumns helpers of starks.cpp:406-555); dst / src point at the first column"""
    _ck(load().xlp_dev_copy_cols(_dp(dst), dst_stride, _dp(src), src_stride, n, dim), "xlp_dev_copy_cols")


def dev_q_split(qq2, qq1, n, n_ext, q_deg, q_dim, shift_in):
    _ck(load().xlp_dev_q_split(_dp(qq2), _dp(qq1), n, n_ext, q_deg, q_dim, shift_in), "xlp_dev_q_split")


def dev_geometric(out, n, dim, first, ratio):
    f = _u64vec(first); r = _u64vec(ratio)
    assert f.size == dim and r.size == dim
    _ck(load().xlp_dev_geometric(_dp(out), n, dim, f.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p)), "xlp_dev_geometric")


def dev_x_div_x_sub_xi(out, x, xi, n):
    v = _u64vec(xi)
    assert v.size == 3
    _ck(load().xlp_dev_x_div_x_sub_xi(_dp(out), _dp(x), v.ctypes.data_as(C.c_void_p), n), "xlp_dev_x_div_x_sub_xi")


class StepsParams(C.Structure):
    """xlp_steps_params (include/xlp_gl.h): device pointers + sizes in elements"""
    _fields_ = [("pols", C.c_void_p), ("pols_size", C.c_uint64),
                ("const_pols", C.c_void_p), ("n_const_pols", C.c_uint64), ("const_pols_size", C.c_uint64),
                ("const_pols_2ns", C.c_void_p), ("n_const_pols_2ns", C.c_uint64), ("const_pols_2ns_size", C.c_uint64),
                ("challenges", C.c_void_p), ("challenges_size", C.c_uint64),
                ("public_inputs", C.c_void_p), ("public_inputs_size", C.c_uint64),
                ("evals", C.c_void_p), ("evals_size", C.c_uint64),
                ("x_n", C.c_void_p), ("x_n_size", C.c_uint64),
                ("x_2ns", C.c_void_p), ("x_2ns_size", C.c_uint64),
                ("zi", C.c_void_p), ("zi_size", C.c_uint64),
                ("x_div_x_sub_xi", C.c_void_p), ("x_div_x_sub_xi_size", C.c_uint64),
                ("x_div_x_sub_wxi", C.c_void_p), ("x_div_x_sub_wxi_size", C.c_uint64),
                ("q_2ns", C.c_void_p), ("q_2ns_size", C.c_uint64),
                ("f_2ns", C.c_void_p), ("f_2ns_size", C.c_uint64),
                ("coset_rows", C.c_uint64), ("coset_stride", C.c_uint64)]


def steps_params(n_const_pols=0, n_const_pols_2ns=0, coset_rows=0, coset_stride=0, **buffers):
    """buffers: name -> device int64 tensor (members of xlp_steps_params); sizes are taken from the tensors"""
    sp = StepsParams()
    sp.n_const_pols, sp.n_const_pols_2ns = n_const_pols, n_const_pols_2ns
    sp.coset_rows, sp.coset_stride = coset_rows, coset_stride
    for name, t in buffers.items():
        setattr(sp, name, t.data_ptr())
        setattr(sp, name + "_size", t.numel())
    sp._keep = buffers
    return sp


class ExprProgram:
    """xlp_expr_program handle (xlp_expr_create / xlp_expr_destroy)"""

    def __init__(self, code, ntmp1, ntmp3, colmajor=False, canonical=False):
        """colmajor: the program addresses column-major buffers (XLP_EXPR_COLMAJOR, include/xlp_gl.h); canonical: and they hold canonical words
        (XLP_EXPR_CANONICAL: what host/starks.hpp promises and the product's plugins are generated under)"""
        code = np.ascontiguousarray(code, dtype=np.uint64)
        self._h = C.c_void_p(None)
        _ck(load().xlp_expr_create_ex(C.byref(self._h), code.ctypes.data_as(C.c_void_p), code.size, ntmp1, ntmp3, (1 if colmajor else 0) | (2 if canonical else 0)), "xlp_expr_create")

    def run(self, params, row0, nrows):
        _ck(load().xlp_dev_expr_run(self._h, C.byref(params), row0, nrows), "xlp_dev_expr_run")

    def codegen(self, path):
        _ck(load().xlp_expr_codegen(self._h, path.encode()), "xlp_expr_codegen")

    def attach(self, so_path):
        _ck(load().xlp_expr_attach(self._h, so_path.encode()), "xlp_expr_attach")

    def compile(self, workdir, name="prog", parts=1):
        """codegen -> hipcc (parts translation units in parallel) -> attach; needs hipcc on the machine"""
        import subprocess
        os.makedirs(workdir, exist_ok=True)
        src, so = os.path.join(workdir, name + ".hip"), os.path.join(workdir, name + ".so")
        os.environ["XLP_EXPR_PARTS"] = str(parts)
        try:
            self.codegen(src)
        finally:
            del os.environ["XLP_EXPR_PARTS"]
        flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I", os.path.join(_HERE, "csrc")] + os.environ.get("XLP_JIT_EXTRA_FLAGS", "").split()
        srcs = [src] + [f"{src}.part{k}.hip" for k in range(parts) if os.path.exists(f"{src}.part{k}.hip")]
        procs = [subprocess.Popen(["hipcc"] + flags + ["-c", f, "-o", f + ".o"]) for f in srcs]
        if any(p.wait() != 0 for p in procs):
            raise XlpError("hipcc failed on a generated expression kernel")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + [f + ".o" for f in srcs])
        self.attach(so)
        return so

    def close(self):
        if self._h:
            load().xlp_expr_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EvmapEntry(C.Structure):
    _fields_ = [("pol", C.c_void_p), ("row_stride", C.c_uint64), ("dim", C.c_uint32), ("prime", C.c_uint32)]


def evmap_entries(entries):
    """entries: list of (section tensor [rows*width] on device, width, column, dim, prime) -> reusable descriptor array"""
    arr = (EvmapEntry * len(entries))()
    for i, (t, width, col, dim, prime) in enumerate(entries):
        arr[i] = EvmapEntry(t.data_ptr() + 8 * col, width, dim, 1 if prime else 0)
    arr._keep = [e[0] for e in entries]
    return arr


def dev_evmap(evals, LEv, LpEv, n, extend_bits, entries):
    """entries: the list evmap_entries() takes, or its result (build it once when the call is repeated)"""
    arr = entries if isinstance(entries, C.Array) else evmap_entries(entries)
    _ck(load().xlp_dev_evmap(_dp(evals), _dp(LEv), _dp(LpEv), n, extend_bits, len(arr), C.cast(arr, C.c_void_p)), "xlp_dev_evmap")


class DotEntry(C.Structure):
    _fields_ = [("pol", C.c_void_p), ("row_stride", C.c_uint64), ("dim", C.c_uint32), ("group", C.c_uint32), ("coef", C.c_uint64)]


def dot_entries(entries):
    """entries: list of (section tensor on device, width, column, dim, group, coefficient index) -> reusable descriptor array"""
    arr = (DotEntry * len(entries))()
    for i, (t, width, col, dim, group, coef) in enumerate(entries):
        arr[i] = DotEntry(t.data_ptr() + 8 * col, width, dim, group, coef)
    arr._keep = [e[0] for e in entries]
    return arr


def dev_dot_columns(out, entries, ngroups, coefs, nrows):
    """out[row][g] = sum of coefs[coef_e] * column_e[row] over the entries of group g (xlp_dev_dot_columns)"""
    arr = entries if isinstance(entries, C.Array) else dot_entries(entries)
    _ck(load().xlp_dev_dot_columns(_dp(out), C.cast(arr, C.c_void_p), len(arr), ngroups, _dp(coefs), coefs.numel(), nrows), "xlp_dev_dot_columns")


def evmap_entries_cm(entries):
    """entries: list of (column-major section tensor, rows per column, column, dim, prime): the columns' first words (coset 0)"""
    arr = (EvmapEntry * len(entries))()
    for i, (t, rows, col, dim, prime) in enumerate(entries):
        arr[i] = EvmapEntry(t.data_ptr() + 8 * col * rows, rows, dim, 1 if prime else 0)
    arr._keep = [e[0] for e in entries]
    return arr


def dev_evmap_cm(evals, LEv, LpEv, n, entries):
    arr = entries if isinstance(entries, C.Array) else evmap_entries_cm(entries)
    _ck(load().xlp_dev_evmap_cm(_dp(evals), _dp(LEv), _dp(LpEv), n, len(arr), C.cast(arr, C.c_void_p)), "xlp_dev_evmap_cm")


def dot_entries_cm(entries):
    """entries: list of (column-major section tensor, rows per column, column, dim, group, coefficient index)"""
    arr = (DotEntry * len(entries))()
    for i, (t, rows, col, dim, group, coef) in enumerate(entries):
        arr[i] = DotEntry(t.data_ptr() + 8 * col * rows, rows, dim, group, coef)
    arr._keep = [e[0] for e in entries]
    return arr


def dev_dot_columns_cm(out, out_cstride, entries, ngroups, coefs, nrows):
    arr = entries if isinstance(entries, C.Array) else dot_entries_cm(entries)
    _ck(load().xlp_dev_dot_columns_cm(_dp(out), out_cstride, C.cast(arr, C.c_void_p), len(arr), ngroups, _dp(coefs), coefs.numel(), nrows), "xlp_dev_dot_columns_cm")


def to_np(t):
    """device int64 tensor -> host uint64 ndarray"""
    return t.detach().cpu().numpy().view(np.uint64)


def from_np(a, device="cuda"):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(device)
