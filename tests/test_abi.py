"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/xlp_gl.h declares, its Python binding mirrors the header, and (on a box without a
GPU) it fails loudly instead of falling back to a CPU path."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    hdr = open(os.path.join(ROOT, "include", "xlp_gl.h")).read()
    return sorted(set(re.findall(r"\b(xlp_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from xlayer_prover_amd import capi
    lib = capi.load()
    names = header_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), f"libxlp_gl.so does not export {n}"


def test_python_binding_covers_header():
    from xlayer_prover_amd import capi
    bound = set(capi.SIGNATURES) | set(capi.VOID_SIGNATURES) | set(capi.STR_SIGNATURES) | set(capi.SIZE_SIGNATURES)
    assert set(header_symbols()) == bound


def test_no_cpu_fallback_without_gpu():
    import torch
    from xlayer_prover_amd import capi
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    import numpy as np
    with pytest.raises(capi.XlpError, match="no HIP device"):
        capi.ntt(np.zeros(8, np.uint64), 8, 1)
    with pytest.raises(capi.XlpError, match="no HIP device"):
        capi.perm([0] * 12)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "xlayer_prover_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "gl_oracle" not in src and "oracle/" not in src.replace("the oracle", ""), f"{f} references the oracle"


def test_field_element_vectors_are_converted_exactly(orc):
    """a list mixing values below and above 2^63 becomes float64 under np.atleast_1d / np.asarray without a dtype and loses its
    low bits (found by the low-degree check of tests/zkevm_checks.py: LEv was built from a rounded xi / shift): the bindings
    convert with an explicit uint64 dtype"""
    import numpy as np
    from xlayer_prover_amd import capi
    P = 0xFFFFFFFF00000001
    v = [2**63 + 5, 3, P - 1]
    assert [int(x) for x in capi._u64vec(v)] == v and [int(x) for x in capi._u64vec(np.array(v, np.uint64))] == v
    assert [int(x) for x in capi._u64vec(7)] == [7]
    g = orc.geometric(4, 3, [1, 0, 0], v)
    assert [int(x) for x in g[1]] == v
    assert [int(x) for x in g[2]] == [int(x) for x in orc.mul3(v, v)]
    x = orc.x_div_x_sub_xi(np.array([11], np.uint64), v)[0]
    den = [orc.sub(11, v[0]), orc.sub(0, v[1]), orc.sub(0, v[2])]
    assert [int(t) for t in orc.mul3(x, den)] == [11, 0, 0]


_COMM_WORKER = r"""
import ctypes, sys
sys.path.insert(0, sys.argv[1])
from xlayer_prover_amd import capi
rank, world, name = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
lib = capi.load()
assert lib.xlp_comm_init(rank, world, b"shm", name.encode()) == 0, lib.xlp_last_error()
assert lib.xlp_comm_rank() == rank and lib.xlp_comm_world() == world
U = ctypes.c_uint64
for it in range(50):                                   # many rounds: the barrier generations and the slots are reused
    words = 1 + it % 7
    mine = (U * words)(*[1000 * rank + 10 * it + k for k in range(words)])
    out = (U * (words * world))()
    assert lib.xlp_comm_all_gather_host(out, mine, words) == 0, lib.xlp_last_error()
    assert list(out) == [1000 * r + 10 * it + k for r in range(world) for k in range(words)], (it, list(out))
    assert lib.xlp_comm_barrier() == 0
lib.xlp_comm_shutdown()
print("ok", rank)
"""


@pytest.mark.parametrize("world", [2, 4])
def test_comm_shared_memory_backend_between_processes(tmp_path, world):
    """the transport's shared-memory backend (csrc/comm.hip) between real processes, no GPU: rendezvous on a named segment, the
    sense-reversing barrier over many generations, the host-word all-gather (what carries Merkle caps, grand-product totals and partial
    evaluations between the ranks of a sharded proof).  The device transfers of the same backend are covered on the GPU by
    tests/test_cpp_sharded.py."""
    import subprocess
    import sys
    w = tmp_path / "worker.py"
    w.write_text(_COMM_WORKER)
    name = f"xlp_abi_{os.getpid()}_{world}"
    procs = [subprocess.Popen([sys.executable, str(w), ROOT, str(r), str(world), name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    try:
        outs = [p.communicate(timeout=120) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and so.strip() == f"ok {r}", se[-1500:]


def test_comm_rejects_bad_worlds():
    from xlayer_prover_amd import capi
    lib = capi.load()
    assert lib.xlp_comm_init(0, 3, b"shm", b"xlp_abi_bad") != 0 and b"power of two" in lib.xlp_last_error()
    assert lib.xlp_comm_init(2, 2, b"shm", b"xlp_abi_bad") != 0
    assert lib.xlp_comm_init(0, 2, None, None) != 0 and b"backend" in lib.xlp_last_error()
    assert lib.xlp_comm_all_gather_host(None, None, 0) != 0 and b"xlp_comm_init" in lib.xlp_last_error()
