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


_COMM_WORKER_FAIL = r"""
import ctypes, os, sys, time
sys.path.insert(0, sys.argv[1])
from xlayer_prover_amd import capi
rank, world, name, mode = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
if rank == 0 and mode == "late0":
    time.sleep(1.0)                                    # the others meet the leftover segment first
lib = capi.load()
assert lib.xlp_comm_init(rank, world, b"shm", name.encode()) == 0, lib.xlp_last_error()
U = ctypes.c_uint64
mine, out = (U * 1)(rank), (U * world)()
assert lib.xlp_comm_all_gather_host(out, mine, 1) == 0, lib.xlp_last_error()
assert list(out) == list(range(world))
if mode in ("abort", "die") and rank == world - 1:
    if mode == "abort":
        lib.xlp_comm_abort(b"boom from the last rank")
    os._exit(3)                                        # no shutdown: the way a crashed rank goes
t0 = time.time()
if mode == "stuck":                                    # a main thread that never returns to the library: the way a rank sits inside ncclRecv / a stream synchronise
    if rank == world - 1:
        os._exit(3)
    time.sleep(60)
    os._exit(4)                                        # not reached: this rank's own watchdog ends the process (exit code 86)
rc = lib.xlp_comm_all_gather_host(out, mine, 1)
if mode in ("abort", "die"):
    assert rc != 0, "the gather must fail when a rank is gone"
    msg = lib.xlp_last_error().decode()
    assert ("boom from the last rank" in msg) if mode == "abort" else ("stopped responding" in msg), msg
    assert time.time() - t0 < 30
    lib.xlp_comm_shutdown()
    print("failed-fast", rank, round(time.time() - t0, 2))
else:
    assert rc == 0, lib.xlp_last_error()
    lib.xlp_comm_shutdown()
    print("ok", rank)
"""


def _run_ranks(tmp_path, script, world, name, mode, env=None, timeout=120):
    import subprocess
    import sys
    w = tmp_path / "worker.py"
    w.write_text(script)
    e = dict(os.environ)
    e.update(env or {})
    procs = [subprocess.Popen([sys.executable, str(w), ROOT, str(r), str(world), name, mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e) for r in range(world)]
    try:
        outs = [p.communicate(timeout=timeout) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return procs, outs


def _leftover_segment(name, world, slot_mb=16, pid=None):
    """what a killed run leaves in /dev/shm: a segment of the right size, marked ready, every rank attached, barrier counters mid-way;
    its creator's pid is dead (or, pid given, a live process that is not the creator: a recycled pid)"""
    import struct
    size = 4096 + world * world * (slot_mb << 20)
    if pid is None:
        pid, start = 2 ** 22 + 12345, 1                  # above pid_max's default: no such process
    else:                                                # a live process with its real start time: the liveness test passes
        stat = open(f"/proc/{pid}/stat").read()
        start = int(stat[stat.rindex(")") + 2:].split()[19])
    hdr = struct.pack("<IIQQQ", 0x584c5031, world, pid, start, size) + struct.pack("<III", 1, 7, world) + b"\0" * 4 + struct.pack("<Q", slot_mb << 20)
    for n, sz, h in ((name, size, hdr), (name + ".ctl", None, None)):
        path = "/dev/shm/" + n
        if h is None:
            if os.path.exists(path):
                os.remove(path)
            continue
        with open(path, "wb") as f:
            f.write(h)
            f.truncate(sz)


@pytest.mark.parametrize("mode", ["ok", "late0", "late0-live-owner"])
def test_comm_rendezvous_ignores_the_segment_a_killed_run_left_behind(tmp_path, mode):
    """a rank > 0 used to accept any existing segment of the right size, and then sat on the stale counters of a dead run
    until the 300 s / 600 s waits gave up.  Now a segment is only taken when its creator is alive, and a rank that did pick up a leftover
    moves over when rank 0 replaces it (here rank 0 arrives a second late)."""
    name = f"xlp_abi_stale_{os.getpid()}_{mode}"
    # "live-owner": the leftover names a live process as its creator (this one), so rank 1 DOES attach to it first -- and must find out
    # that the name has moved to rank 0's fresh segment
    _leftover_segment(name, 2, pid=os.getpid() if mode.endswith("live-owner") else None)
    try:
        procs, outs = _run_ranks(tmp_path, _COMM_WORKER_FAIL, 2, name, mode.split("-")[0])
        for r, (p, (so, se)) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and so.strip() == f"ok {r}", se[-1500:]
    finally:
        for n in (name, name + ".ctl"):
            if os.path.exists("/dev/shm/" + n):
                os.remove("/dev/shm/" + n)


@pytest.mark.parametrize("mode", ["abort", "die"])
def test_comm_failure_of_one_rank_ends_the_others(tmp_path, mode):
    """a dead rank used to leave the others in a 600 s barrier wait.  xlp_comm_abort (the host's exit path calls it) raises
    a flag in the control segment that every wait checks; a rank that just dies is found by its heartbeat (shortened here)."""
    name = f"xlp_abi_fail_{os.getpid()}_{mode}"
    try:
        procs, outs = _run_ranks(tmp_path, _COMM_WORKER_FAIL, 2, name, mode, env={"XLP_COMM_HEARTBEAT_STALE_MS": "1500"})
        assert procs[1].returncode == 3
        assert procs[0].returncode == 0 and outs[0][0].startswith("failed-fast 0"), outs[0][1][-1500:]
    finally:
        for n in (name, name + ".ctl"):
            if os.path.exists("/dev/shm/" + n):
                os.remove("/dev/shm/" + n)


def test_comm_watchdog_that_finds_the_dead_peer_itself_still_ends_its_process(tmp_path):
    """the watchdog skipped the kill whenever the abort flag carried its own rank -- also when IT had raised the flag for a peer whose
    heartbeat went stale, so with two ranks the survivor sat in its RCCL call for ever.  Who raised the flag is now separate from who failed: only
    a flag raised by this rank's MAIN thread (already on its way out) spares the process.  (The kill is the RCCL transport's; the hook runs it over shm.)"""
    name = f"xlp_abi_stuck_{os.getpid()}"
    try:
        procs, outs = _run_ranks(tmp_path, _COMM_WORKER_FAIL, 2, name, "stuck", env={"XLP_COMM_HEARTBEAT_STALE_MS": "1000", "XLP_COMM_WATCHDOG_ENDS_PROCESS": "1"}, timeout=50)
        assert procs[1].returncode == 3
        assert procs[0].returncode == 86 and "watchdog gave up on a peer" in outs[0][1] and "rank 1 stopped responding" in outs[0][1], (procs[0].returncode, outs[0][1][-1500:])
    finally:
        for n in (name, name + ".ctl"):
            if os.path.exists("/dev/shm/" + n):
                os.remove("/dev/shm/" + n)


def test_comm_rejects_bad_worlds():
    from xlayer_prover_amd import capi
    lib = capi.load()
    assert lib.xlp_comm_init(0, 3, b"shm", b"xlp_abi_bad") != 0 and b"power of two" in lib.xlp_last_error()
    assert lib.xlp_comm_init(2, 2, b"shm", b"xlp_abi_bad") != 0
    assert lib.xlp_comm_init(0, 2, None, None) != 0 and b"backend" in lib.xlp_last_error()
    assert lib.xlp_comm_all_gather_host(None, None, 0) != 0 and b"xlp_comm_init" in lib.xlp_last_error()
