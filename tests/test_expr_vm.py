"""Constraint-expression VM (csrc/expr.hip) against the oracle interpreter on seeded random
programs that use every operand kind, op and dimension combination, plus the rejection paths."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN_PROOFS

from tests.expr_programs import accumulation_program, long_chain_program, make_buffers, random_program
from xlayer_prover_amd import expr as E


@pytest.fixture(scope="module")
def xlp():
    import torch
    from xlayer_prover_amd import capi
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.init(0)
    capi.use_torch_stream()
    return capi


def _run_both(xlp, orc, prog, bufs, row0, nrows, nconst, nconst2):
    dev = {k: xlp.from_np(v) for k, v in bufs.items()}
    sp = xlp.steps_params(n_const_pols=nconst, n_const_pols_2ns=nconst2, **dev)
    p = xlp.ExprProgram(prog.code, prog.ntmp1, prog.ntmp3)
    p.run(sp, row0, nrows); xlp.sync()
    p.close()
    exp = {k: v.copy() for k, v in bufs.items()}
    orc.expr_run(prog.code, prog.ntmp1, prog.ntmp3, exp, row0, nrows, nconst, nconst2)
    for k in ("pols", "q_2ns", "f_2ns"):
        got = xlp.to_np(dev[k])
        assert (got == exp[k]).all(), f"{k}: {np.flatnonzero(got != exp[k])[:5]}"
    return exp


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,ninstr,row0,nrows", [(1, 64, 40, 0, 64), (2, 1000, 300, 0, 1000), (3, 1 << 12, 500, 0, 1 << 12),
                                                      (4, 777, 200, 100, 500), (5, 1 << 16, 250, 0, 1 << 16), (6, 5, 120, 0, 5), (7, 4096, 150, 4095, 1)])
def test_expr_random_programs_vs_oracle(xlp, orc, seed, n, ninstr, row0, nrows):
    prog, width = random_program(seed, n, ninstr)
    bufs = make_buffers(orc, 0x5EED0100 + seed, n, width)
    exp = _run_both(xlp, orc, prog, bufs, row0, nrows, 5, 4)
    assert (exp["pols"] != bufs["pols"]).any()          # the program stored something


@pytest.mark.gpu
def test_expr_many_temporaries_and_full_occupancy(xlp, orc):
    """zkEVM-scale register file: step42ns needs 1196 + 3 * 175 slots per row (zkevm.chelpers.step42ns.parser.hpp:3-4)"""
    n = 1 << 17
    prog, width = random_program(11, n, 1500, ntmp1=1196, ntmp3=175, nstore=20)
    bufs = make_buffers(orc, 0x5EED0111, n, width)
    _run_both(xlp, orc, prog, bufs, 0, n, 5, 4)


@pytest.mark.gpu
def test_expr_rejections(xlp, orc):
    ok = E.Program().add(E.tmp1(0), E.number(1), E.number(2))
    c = ok.code.copy(); c[0] = (c[0] & ~np.uint64(0xff)) | np.uint64(7)
    with pytest.raises(xlp.XlpError, match="unknown op"):
        xlp.ExprProgram(c, 1, 0)
    with pytest.raises(xlp.XlpError, match="tmp1 index out of range"):
        xlp.ExprProgram(ok.code, 0, 0)
    c = ok.code.copy(); c[0] = (c[0] & ~np.uint64(0xff00)) | np.uint64(E.K_NUMBER << 8)
    with pytest.raises(xlp.XlpError, match="cannot be a destination"):
        xlp.ExprProgram(c, 1, 0)
    c = E.Program().add(E.tmp3(0), E.number(1), E.challenge(0)).code
    c[0] = (c[0] & ~np.uint64(3 << 32)) | np.uint64(1 << 32)
    with pytest.raises(xlp.XlpError, match="dimension"):
        xlp.ExprProgram(c, 0, 1)
    with pytest.raises(xlp.XlpError, match="truncated"):
        xlp.ExprProgram(ok.code[:2], 1, 0)
    # bounds: pols too short for the last row, missing buffer, column beyond nConstPols
    n, w = 16, 4
    prog = E.Program().copy(E.pols(3, w), E.pols(0, w))
    p = xlp.ExprProgram(prog.code, 0, 0)
    full = xlp.from_np(np.zeros(n * w, np.uint64))
    p.run(xlp.steps_params(pols=full), 0, n)
    with pytest.raises(xlp.XlpError, match="past the end"):
        p.run(xlp.steps_params(pols=full[: n * w - 1]), 0, n)
    with pytest.raises(xlp.XlpError, match="past the end"):
        p.run(xlp.steps_params(pols=full), 1, n)
    with pytest.raises(xlp.XlpError, match="no buffer"):
        p.run(xlp.steps_params(), 0, n)
    p2 = xlp.ExprProgram(E.Program().copy(E.tmp1(0), E.const(5)).code, 1, 0)
    with pytest.raises(xlp.XlpError, match="past the end"):
        p2.run(xlp.steps_params(n_const_pols=5, const_pols=xlp.from_np(np.zeros(5 * n, np.uint64))), 0, n)
    xlp.sync()


def test_expr_assembler_encoding():
    prog = E.Program().mul(E.tmp3(2), E.pols_prime(7, 1, 64, 9, 3), E.tmp1(4))
    c = prog.code
    assert c.size == 1 + 1 + 4 + 1 and prog.ntmp1 == 5 and prog.ntmp3 == 3
    assert int(c[0]) == E.MUL | E.K_TMP3 << 8 | E.K_POLS_PRIME << 16 | E.K_TMP1 << 24 | 3 << 32 | 3 << 34 | 1 << 36
    assert [int(v) for v in c[1:]] == [2, 7, 1, 64, 9, 4]
    with pytest.raises(ValueError):
        E.Operand(E.K_POLS, (1,), 1)


REAL_STEPS = ["step2prev", "step3prev", "step3", "step42ns", "step52ns"]
OTHER_PROGRAMS = [f"{pil}.{step}" for pil in ("c12a", "recursive1", "recursive2", "recursivef") for step in ("step3prev", "step3", "step42ns", "step52ns")]


@pytest.mark.gpu
@pytest.mark.parametrize("name", OTHER_PROGRAMS)
def test_expr_reference_recursive_programs(xlp, orc, name):
    """The constraint programs of the reference's other STARKs (c12a, recursive1, recursive2, recursiveF), converted
    from their straight-line generated C++ by tools/chelpers_cpp_to_xlp.py at build time: HIP VM == oracle, 256 rows."""
    from tests import expr_real as R
    path = os.path.join(R.REF_DIR, f"{name}.xlpexpr")
    if not os.path.exists(path):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    code, ntmp1, ntmp3 = R.read_program(path)
    n = 256
    code, info = R.shrink(code, n)
    bufs = R.buffers(orc.fill, 0x5EED0A00, n, info)
    dev = {k: xlp.from_np(v) for k, v in bufs.items()}
    p = xlp.ExprProgram(code, ntmp1, ntmp3)
    p.run(xlp.steps_params(n_const_pols=info["n_const"], n_const_pols_2ns=info["n_const2"], **dev), 0, n); xlp.sync(); p.close()
    exp = {k: v.copy() for k, v in bufs.items()}
    orc.expr_run(code, ntmp1, ntmp3, exp, 0, n, info["n_const"], info["n_const2"])
    changed = 0
    for k in ("pols", "q_2ns", "f_2ns"):
        assert (xlp.to_np(dev[k]) == exp[k]).all(), k
        changed += int((exp[k] != bufs[k]).sum())
    assert changed > 0


@pytest.mark.gpu
@pytest.mark.parametrize("step", REAL_STEPS)
def test_expr_reference_zkevm_programs(xlp, orc, step):
    """The reference's checked-in zkEVM (fork 9) constraint programs, converted by tools/chelpers_to_xlp.py at
    build time (xlayer_prover_amd/chelpers, not in the repo), on a 256-row synthetic trace: HIP VM == oracle,
    interpreted and compiled."""
    import os
    from tests import expr_real as R
    path = os.path.join(R.REF_DIR, f"zkevm.{step}.xlpexpr")
    if not os.path.exists(path):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    code, ntmp1, ntmp3 = R.read_program(path)
    n = 256
    code, info = R.shrink(code, n)
    bufs = R.buffers(orc.fill, 0x5EED0200, n, info)
    dev = {k: xlp.from_np(v) for k, v in bufs.items()}
    sp = xlp.steps_params(n_const_pols=info["n_const"], n_const_pols_2ns=info["n_const2"], **dev)
    p = xlp.ExprProgram(code, ntmp1, ntmp3)
    p.run(sp, 0, n); xlp.sync()
    exp = {k: v.copy() for k, v in bufs.items()}
    orc.expr_run(code, ntmp1, ntmp3, exp, 0, n, info["n_const"], info["n_const2"])
    changed = 0
    for k in ("pols", "q_2ns", "f_2ns"):
        assert (xlp.to_np(dev[k]) == exp[k]).all(), k
        changed += int((exp[k] != bufs[k]).sum())
    assert changed > 0
    # the same program compiled ahead of time (tools/build_expr_plugins.py, built where hipcc + the reference are)
    so = os.path.join(R.REF_DIR, f"zkevm.{step}.n8.so")
    if os.path.exists(so):
        dev2 = {k: xlp.from_np(v) for k, v in bufs.items()}
        sp2 = xlp.steps_params(n_const_pols=info["n_const"], n_const_pols_2ns=info["n_const2"], **dev2)
        p.attach(so)
        p.run(sp2, 0, n); xlp.sync()
        for k in ("pols", "q_2ns", "f_2ns"):
            assert (xlp.to_np(dev2[k]) == exp[k]).all(), "compiled " + k
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,ninstr,row0,nrows", [(21, 512, 120, 0, 512), (22, 1000, 400, 37, 900), (23, 1 << 14, 700, 0, 1 << 14)])
def test_expr_compiled_random_programs_vs_oracle(xlp, orc, tmp_path, seed, n, ninstr, row0, nrows):
    """codegen -> hipcc -> attach on the GPU box, then the compiled kernels against the oracle (and a plugin
    generated from another program is refused)"""
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    prog, width = random_program(seed, n, ninstr)
    bufs = make_buffers(orc, 0x5EED0300 + seed, n, width)
    dev = {k: xlp.from_np(v) for k, v in bufs.items()}
    sp = xlp.steps_params(n_const_pols=5, n_const_pols_2ns=4, **dev)
    p = xlp.ExprProgram(prog.code, prog.ntmp1, prog.ntmp3)
    so = p.compile(str(tmp_path), "rnd", parts=4)
    p.run(sp, row0, nrows); xlp.sync()
    exp = {k: v.copy() for k, v in bufs.items()}
    orc.expr_run(prog.code, prog.ntmp1, prog.ntmp3, exp, row0, nrows, 5, 4)
    for k in ("pols", "q_2ns", "f_2ns"):
        got = xlp.to_np(dev[k])
        assert (got == exp[k]).all(), f"{k}: {np.flatnonzero(got != exp[k])[:5]}"
    other, _ = random_program(seed + 100, n, 50)
    q = xlp.ExprProgram(other.code, other.ntmp1, other.ntmp3)
    with pytest.raises(xlp.XlpError, match="different program"):
        q.attach(so)
    p.close(); q.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,nterms", [(71, 1 << 12, 300), (72, 777, 1300)])
def test_expr_compiled_lazy_accumulation_vs_oracle(xlp, orc, tmp_path, seed, n, nterms):
    """the generator's carry-free accumulators (gl::lazy3: mac, mac3, neg_any, fold) compiled by hipcc and run on the GPU against the oracle:
    accumulation chains with hops, stale states, copies, Fp^3 values; all-ones rows and non-canonical challenge words; the second case puts
    ONE chain of 1 300 products with every coefficient limb and the first row's values maximal into one segment (the accumulator is folded and restarted
    after 1 000 products: the sums stay below 2^64)"""
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    if nterms > 1000:
        os.environ["XLP_EXPR_SEG"] = "100000"
    try:
        prog, width = long_chain_program(seed, n, nterms) if nterms > 1000 else accumulation_program(seed, n, nterms)
        bufs = make_buffers(orc, 0x5EED0800 + seed, n, width)
        if nterms > 1000:
            bufs["challenges"][:] = 2**64 - 1
        bufs["challenges"][:3] = [2**64 - 1, 0xFFFFFFFF00000001, 0xFFFFFFFF00000006]
        bufs["pols"][:width] = 2**64 - 1
        dev = {k: xlp.from_np(v) for k, v in bufs.items()}
        sp = xlp.steps_params(n_const_pols=5, n_const_pols_2ns=4, **dev)
        p = xlp.ExprProgram(prog.code, prog.ntmp1, prog.ntmp3)
        p.compile(str(tmp_path), "lazy", parts=4)
    finally:
        os.environ.pop("XLP_EXPR_SEG", None)
    text = open(tmp_path / "lazy.hip").read() + "".join(open(f).read() for f in sorted(map(str, tmp_path.glob("lazy.hip.part*.hip"))))
    assert ".mac(" in text and ".fold(" in text and (nterms > 1000 or ".mac3(" in text)
    p.run(sp, 0, n); xlp.sync(); p.close()
    exp = {k: v.copy() for k, v in bufs.items()}
    orc.expr_run(prog.code, prog.ntmp1, prog.ntmp3, exp, 0, n, 5, 4)
    for k in ("pols", "q_2ns", "f_2ns"):
        got = xlp.to_np(dev[k])
        assert (got == exp[k]).all(), f"{k}: {np.flatnonzero(got != exp[k])[:5]}"


def test_converter_covers_reference_tables():
    """Every case of the five zkEVM parsers is recognised and the op/arg tables are consumed exactly
    (NOPS_ / NARGS_); runs only where the reference tree is mounted."""
    import os
    import sys
    ref = "/root/reference/src/starkpil/zkevm/chelpers"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import chelpers_to_xlp as conv
    expect = {"step2prev": (2094, 6804), "step3prev": (6673, 21766), "step3": (13117, 55745), "step42ns": (18546, 94863), "step52ns": (3001, 7619)}
    for step, (nops, nargs) in expect.items():
        prog, ntmp1, ntmp3, info = conv.convert(f"{ref}/zkevm.chelpers.{step}.parser.cpp", f"{ref}/zkevm.chelpers.{step}.parser.hpp", step)
        assert (info["nops"], info["nargs"]) == (nops, nargs) and info["ninstr"] >= nops


def _run_generated_on_cpu(tmp_path, code, ntmp1, ntmp3, bufs, row0, nrows, nconst, nconst2, hoist=0):
    """codegen (no GPU needed) -> host compiler with the CPU stand-ins of tests/jit_cpu + ASan -> run.  hoist: the generator pipelines the loads
    in batches of this many (XLP_EXPR_HOIST_LOADS, csrc/expr.hip place_loads: how the product's plugins are built)"""
    import os
    import subprocess
    from xlayer_prover_amd import capi
    here = os.path.dirname(os.path.abspath(__file__))
    p = capi.ExprProgram(code, ntmp1, ntmp3)
    src = str(tmp_path / "prog.hip")
    if hoist:
        os.environ["XLP_EXPR_HOIST_LOADS"] = str(hoist)
    try:
        p.codegen(src); p.close()
    finally:
        os.environ.pop("XLP_EXPR_HOIST_LOADS", None)
    exe = str(tmp_path / "harness")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17", "-x", "c++",
                           f"-DPLUGIN_SRC=\"{src}\"", "-I", os.path.join(here, "jit_cpu"), "-I", os.path.join(here, "..", "xlayer_prover_amd", "csrc"),
                           os.path.join(here, "jit_cpu", "harness.cpp"), "-o", exe])
    order = ["pols", "const_pols", "const_pols_2ns", "challenges", "public_inputs", "evals", "x_n", "x_2ns", "zi", "x_div_x_sub_xi",
             "x_div_x_sub_wxi", "q_2ns", "f_2ns"]
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.array([row0, nrows, nconst, nconst2, 0, 0, 0, 0], np.uint64).tobytes())
        for k in order:
            f.write(np.array([bufs[k].size], np.uint64).tobytes()); f.write(bufs[k].tobytes())
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    raw = np.fromfile(tmp_path / "out.bin", dtype=np.uint64)
    out, q = {}, 0
    for k in ("pols", "q_2ns", "f_2ns"):
        n = int(raw[q]); out[k] = raw[q + 1:q + 1 + n]; q += 1 + n
    return out


@pytest.mark.parametrize("seed,n,ninstr,row0,nrows", [(31, 64, 150, 0, 64), (32, 300, 700, 11, 250)])
def test_expr_generated_code_on_cpu(orc, tmp_path, seed, n, ninstr, row0, nrows):
    """The ahead-of-time generator's output (same text hipcc compiles) against the oracle, on the CPU under
    AddressSanitizer: segmenting, liveness across segments, value forwarding and the staged-column indexing."""
    os.environ["XLP_EXPR_SEG"] = "64"
    try:
        prog, width = random_program(seed, n, ninstr)
        bufs = make_buffers(orc, 0x5EED0400 + seed, n, width)
        got = _run_generated_on_cpu(tmp_path, prog.code, prog.ntmp1, prog.ntmp3, bufs, row0, nrows, 5, 4)
    finally:
        del os.environ["XLP_EXPR_SEG"]
    exp = {k: v.copy() for k, v in bufs.items()}
    orc.expr_run(prog.code, prog.ntmp1, prog.ntmp3, exp, row0, nrows, 5, 4)
    for k in ("pols", "q_2ns", "f_2ns"):
        assert (got[k] == exp[k]).all(), f"{k}: {np.flatnonzero(got[k] != exp[k])[:5]}"


@pytest.mark.parametrize("seed,n,ninstr,seg,batch,cut", [(33, 128, 700, 256, 4, 0), (35, 300, 700, 128, 3, 0), (37, 128, 900, 0, 4, 160)])
def test_expr_generated_code_with_pipelined_loads_on_cpu(orc, tmp_path, seed, n, ninstr, seg, batch, cut):
    """the generator's load pipelining (the loads of a segment issued a batch ahead of their use: how the product's plugins are built), also with
    the segment boundaries chosen by XLP_EXPR_CUT: the text computes what the oracle computes, and loads did move"""
    if seg:
        os.environ["XLP_EXPR_SEG"] = str(seg)
    if cut:
        os.environ["XLP_EXPR_CUT"] = str(cut); os.environ["XLP_EXPR_CUT_MIN"] = "16"
    try:
        prog, width = random_program(seed, n, ninstr)
        bufs = make_buffers(orc, 0x5EED0900 + seed, n, width)
        got = _run_generated_on_cpu(tmp_path, prog.code, prog.ntmp1, prog.ntmp3, bufs, 0, n, 5, 4, hoist=batch)
    finally:
        for k in ("XLP_EXPR_SEG", "XLP_EXPR_CUT", "XLP_EXPR_CUT_MIN"):
            os.environ.pop(k, None)
    text = open(tmp_path / "prog.hip").read()
    assert "_raw = " in text and "__builtin_amdgcn_sched_barrier(0);" in text
    exp = {k: v.copy() for k, v in bufs.items()}
    orc.expr_run(prog.code, prog.ntmp1, prog.ntmp3, exp, 0, n, 5, 4)
    for k in ("pols", "q_2ns", "f_2ns"):
        assert (got[k] == exp[k]).all(), f"{k}: {np.flatnonzero(got[k] != exp[k])[:5]}"


@pytest.mark.parametrize("kind,seed,n,size,seg", [("random", 41, 64, 700, 128), ("random", 42, 64, 1500, 0), ("accumulation", 43, 64, 400, 96), ("random", 44, 96, 900, 64)])
def test_expr_generated_representations_on_small_values_on_cpu(orc, tmp_path, kind, seed, n, size, seg):
    """"any" words where canonical ones are required (csrc/expr.hip, REPRESENTATIONS: a product is only canonicalised where its reader needs it): the CPU
    stand-in of gl_field.hpp traps on a non-canonical word handed to addc / subc / mul3 / mac3 / neg / an accumulator base, and its mul / add / sub return the
    non-canonical representative whenever r + p fits 64 bits -- which needs SMALL values, so every input here is one of 0, 1, 2, 3, p - 1, p - 2, p, p + 1
    (products and sums of those stay small for many instructions).  Outputs equal the oracle's: what is stored is canonical."""
    P = 0xFFFFFFFF00000001
    prog, width = (random_program(seed, n, size) if kind == "random" else accumulation_program(seed, n, size))
    bufs = make_buffers(orc, 0x5EED0A00 + seed, n, width)
    pick = np.array([0, 1, 2, 3, P - 1, P - 2, P, P + 1], dtype=np.uint64)
    rng = np.random.default_rng(seed)
    for k in bufs:
        bufs[k] = pick[rng.integers(0, len(pick) if k == "pols" else 6, size=bufs[k].size)]      # (only pols may hold non-canonical words)
    env = {"XLP_EXPR_SEG": str(seg)} if seg else {"XLP_EXPR_CUT": "200", "XLP_EXPR_CUT_MIN": "16"}
    os.environ.update(env)
    try:
        got = _run_generated_on_cpu(tmp_path, prog.code, prog.ntmp1, prog.ntmp3, bufs, 0, n, 5, 4, hoist=3)
    finally:
        for k in env:
            del os.environ[k]
    text = open(tmp_path / "prog.hip").read()
    assert "gl::mul(" in text and ("gl::add(" in text or "gl::sub(" in text) and "gl::canon(v" in text
    exp = {k: v.copy() for k, v in bufs.items()}
    orc.expr_run(prog.code, prog.ntmp1, prog.ntmp3, exp, 0, n, 5, 4)
    for k in ("pols", "q_2ns", "f_2ns"):
        assert (got[k] == exp[k]).all(), f"{k}: {np.flatnonzero(got[k] != exp[k])[:5]}"


def _codegen_text(tmp_path, prog, name, **env):
    from xlayer_prover_amd import capi
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        p = capi.ExprProgram(prog.code, prog.ntmp1, prog.ntmp3)
        src = str(tmp_path / name)
        p.codegen(src); p.close()
    finally:
        for k in env:
            del os.environ[k]
    return open(src).read()


def test_expr_load_pipelining_is_a_pure_reordering_of_the_generated_text(tmp_path):
    """undoing the pipelining line by line -- the raw load back in front of its use, the fences dropped -- gives the text the generator writes
    without it: nothing is added, lost or altered but the place (and the name) of a load.  A load of a column the segment also WRITES is never moved."""
    import re
    prog, _ = random_program(36, 256, 1200)
    text = _codegen_text(tmp_path, prog, "plain.hip", XLP_EXPR_SEG=200)
    out = _codegen_text(tmp_path, prog, "piped.hip", XLP_EXPR_SEG=200, XLP_EXPR_HOIST_LOADS=5)
    assert out != text and out.count("_raw = ") >= 10
    assert not re.search(r"_raw = (CW|PW)\[", out)                                      # written columns are read in place, through the writing pointer
    raw = {}
    back = []
    for l in out.split("\n"):
        m = re.match(r"^(\s*)const u64 (v\d+)_raw = (.*);$", l)
        if m:
            assert m.group(2) not in raw                                  # (names are per kernel: cleared at each kernel head below)
            raw[m.group(2)] = m.group(3)
            continue
        if l.strip() == "__builtin_amdgcn_sched_barrier(0);":
            continue
        if l.startswith('extern "C" __global__'):
            raw = {}
        m = re.match(r"^(\s*)const u64 (v\d+) = (gl::canon\()?(v\d+)_raw\)?;$", l)
        if m:
            assert m.group(2) == m.group(4) and m.group(2) in raw         # defined earlier in this kernel
            l = f"{m.group(1)}const u64 {m.group(2)} = " + (f"gl::canon({raw[m.group(2)]});" if m.group(3) else f"{raw[m.group(2)]};")
        back.append(l)
    strip = lambda t: "\n".join(x for x in t.split("\n") if not x.startswith("// xlp_expr_codegen:"))
    assert strip("\n".join(back)) == strip(text)
    # one batch ahead: the raw loads of batch b + 1 stand in front of the first use of batch b
    k = out[out.index('seg_0('):]
    names = re.findall(r"const u64 (v\d+)_raw = ", k[:k.index("\n}\n")])
    assert len(names) >= 15 and k.index(f"const u64 {names[5]} = ") > k.index(f"const u64 {names[9]}_raw = ") and k.index(f"const u64 {names[4]} = ") < k.index(f"const u64 {names[10]}_raw = ")


def test_expr_segment_cuts_by_dynamic_programming_move_fewer_words(tmp_path):
    """XLP_EXPR_CUT: the generator's own account (the stats line of the generated file) of the words a row moves through HBM is not larger with the
    boundaries chosen by dynamic programming than with a cut every so many instructions of the same maximum length, and no segment is longer"""
    import re
    prog, _ = random_program(38, 256, 3000, ntmp1=40, ntmp3=20)
    def words(text):
        m = re.search(r"// xlp_expr_codegen: (\d+) instructions in (\d+) segments \(longest (\d+)\), per row (\d+) column words loaded, (\d+) temporary words loaded, (\d+) temporary words stored", text)
        return int(m.group(4)) + int(m.group(5)) + int(m.group(6)), int(m.group(3)), int(m.group(2))
    fixed = words(_codegen_text(tmp_path, prog, "fixed.hip", XLP_EXPR_SEG=192))
    dp = words(_codegen_text(tmp_path, prog, "dp.hip", XLP_EXPR_CUT=192, XLP_EXPR_CUT_MIN=32, XLP_EXPR_CUT_SEGCOST=0))
    assert dp[1] <= 192 and dp[0] <= fixed[0], (fixed, dp)


@pytest.mark.parametrize("seed,n,nterms,seg,accs", [(61, 64, 120, 64, 3), (62, 100, 400, 256, 3), (63, 64, 1500, 100000, 1), (64, 64, 200, 32, 2), (65, 64, 1300, 100000, -1)])
def test_expr_generated_lazy_accumulation_on_cpu(orc, tmp_path, seed, n, nterms, seg, accs):
    """The generator's carry-free accumulators (acc += challenge x value as gl::lazy3::mac, csrc/expr.hip) against the oracle, on the
    CPU stand-in that traps on any 64-bit overflow of a partial sum: chains that hop between temporaries, states read after the
    accumulator moved on, more chains than accumulators, one segment of 1 500 terms (past the 1 000 an accumulator is given), and
    challenge words that are not canonical."""
    os.environ["XLP_EXPR_SEG"] = str(seg); os.environ["XLP_EXPR_LAZY_ACCS"] = str(accs)
    try:
        long_chain = accs < 0                                                             # accs = -1: ONE chain of nterms products (the 1 000-product limit)
        if long_chain:
            os.environ["XLP_EXPR_LAZY_ACCS"] = "3"
        prog, width = long_chain_program(seed, n, nterms) if long_chain else accumulation_program(seed, n, nterms)
        bufs = make_buffers(orc, 0x5EED0700 + seed, n, width)
        bufs["challenges"][:] = 2**64 - 1 if long_chain else bufs["challenges"]          # every coefficient limb maximal
        bufs["challenges"][:3] = [2**64 - 1, 0xFFFFFFFF00000001, 0xFFFFFFFF00000006]     # any 64-bit word
        bufs["pols"][:width] = 2**64 - 1                                                  # row 0: every value all ones
        got = _run_generated_on_cpu(tmp_path, prog.code, prog.ntmp1, prog.ntmp3, bufs, 0, n, 5, 4)
        text = open(tmp_path / "prog.hip").read()
        if long_chain:
            assert text.count(".mac(") == nterms and text.count("gl::lazy3") == 2 and text.count(".fold(") == 2 and "neg_any" in text
        else:
            assert ".mac(" in text and ".mac3(" in text and ".fold(" in text and "neg_any" in text
    finally:
        del os.environ["XLP_EXPR_SEG"]; del os.environ["XLP_EXPR_LAZY_ACCS"]
    exp = {k: v.copy() for k, v in bufs.items()}
    orc.expr_run(prog.code, prog.ntmp1, prog.ntmp3, exp, 0, n, 5, 4)
    for k in ("pols", "q_2ns", "f_2ns"):
        assert (got[k] == exp[k]).all(), f"{k}: {np.flatnonzero(got[k] != exp[k])[:5]}"


@pytest.mark.parametrize("step", ["step2prev"])
def test_expr_generated_code_on_cpu_reference_programs(orc, tmp_path, step):
    """same CPU check on two of the converted reference programs (64-row synthetic slab)"""
    from tests import expr_real as R
    path = os.path.join(R.REF_DIR, f"zkevm.{step}.xlpexpr")
    if not os.path.exists(path):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    code, ntmp1, ntmp3 = R.read_program(path)
    n = 64
    code, info = R.shrink(code, n)
    bufs = R.buffers(orc.fill, 0x5EED0500, n, info)
    got = _run_generated_on_cpu(tmp_path, code, ntmp1, ntmp3, bufs, 0, n, info["n_const"], info["n_const2"])
    exp = {k: v.copy() for k, v in bufs.items()}
    orc.expr_run(code, ntmp1, ntmp3, exp, 0, n, info["n_const"], info["n_const2"])
    for k in ("pols", "q_2ns", "f_2ns"):
        assert (got[k] == exp[k]).all(), k


@pytest.mark.gpu
@pytest.mark.parametrize("compiled", [False, True])
def test_expr_multi_slab_run(xlp, orc, tmp_path, compiled):
    """More rows than one launch holds (2^19-row slabs on a 256-CU part): slab seams, halo rows of next-row operands
    and the wrap-around at the end of the domain, checked on row windows against the oracle."""
    import shutil
    if compiled and shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    n = (1 << 20) + 4096
    prog, width = random_program(41, n, 160, nstore=8)
    bufs = make_buffers(orc, 0x5EED0600, n, width)
    dev = {k: xlp.from_np(v) for k, v in bufs.items()}
    sp = xlp.steps_params(n_const_pols=5, n_const_pols_2ns=4, **dev)
    p = xlp.ExprProgram(prog.code, prog.ntmp1, prog.ntmp3)
    if compiled:
        p.compile(str(tmp_path), "slab", parts=2)
    p.run(sp, 0, n); xlp.sync(); p.close()
    got = {k: xlp.to_np(dev[k]) for k in ("pols", "q_2ns", "f_2ns")}
    slab = 1 << 19
    for lo, cnt in [(0, 48), (slab - 40, 80), (2 * slab - 40, 80), (n - 48, 48)]:
        exp = {k: v.copy() for k, v in bufs.items()}
        orc.expr_run(prog.code, prog.ntmp1, prog.ntmp3, exp, lo, cnt, 5, 4)
        # rows the window wrote: same-row stores land in [lo, lo+cnt); next-row stores up to 2 rows further (mod n)
        rows = np.arange(lo, lo + cnt)
        for k, per_row in (("pols", width), ("q_2ns", 3), ("f_2ns", 3)):
            e = exp[k].reshape(n, per_row); b = bufs[k].reshape(n, per_row); g = got[k].reshape(n, per_row)
            touched = (e != b)
            assert (g[touched] == e[touched]).all(), (k, lo)
            assert touched[rows].any() or k != "pols"


@pytest.mark.parametrize("step,cpp,fn", [("step2prev", "step2", "step2prev_first"), ("step3prev", "step3prev", "step3prev_first"),
                                         ("step52ns", "step52ns", "step52ns_first")])
def test_reference_table_and_straight_line_forms_agree(orc, step, cpp, fn):
    """The reference ships three zkEVM steps in TWO generated forms: interpreter tables (`*.parser.hpp` + the case
    switch of `*.parser.cpp`) and straight-line C++ (`zkevm.chelpers.step3prev.cpp` ...).  Converted independently
    (tools/chelpers_to_xlp.py / tools/chelpers_cpp_to_xlp.py) and run by the oracle on one synthetic trace they must
    write identical outputs, instruction for instruction count included: this pins the reading of the table format
    (operand order of every case, fused cases, the accumulator cases of step52ns) to a second reference artefact."""
    import sys
    ref = "/root/reference/src/starkpil/zkevm/chelpers"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import chelpers_cpp_to_xlp as conv2
    import chelpers_to_xlp as conv1
    from tests import expr_real as R
    p1, ntmp1a, ntmp3a, info1 = conv1.convert(f"{ref}/zkevm.chelpers.{step}.parser.cpp", f"{ref}/zkevm.chelpers.{step}.parser.hpp", step)
    p2 = conv2.convert(f"{ref}/zkevm.chelpers.{cpp}.cpp", fn)
    assert p1.ninstr == p2.ninstr and len(p1.words) == len(p2.words)
    n = 48
    (c1, c2), info = R.shrink_many([p1.code, p2.code], n)
    bufs = R.buffers(orc.fill, 0x5EED0900, n, info)
    out1 = {k: v.copy() for k, v in bufs.items()}
    out2 = {k: v.copy() for k, v in bufs.items()}
    orc.expr_run(c1, ntmp1a, ntmp3a, out1, 0, n, info["n_const"], info["n_const2"])
    orc.expr_run(c2, p2.ntmp1, p2.ntmp3, out2, 0, n, info["n_const"], info["n_const2"])
    changed = 0
    for k in ("pols", "q_2ns", "f_2ns"):
        assert (out1[k] == out2[k]).all(), k
        changed += int((out1[k] != bufs[k]).sum())
    assert changed > 100


def _golden_fri_polynomial_inputs(orc, golden_dir, proof_name, prog_name):
    """For every query of a golden recursive proof: a one-row trace made of the OPENED VALUES (rows of trees 1, 3, 4
    and the constant tree), the proof's evaluations and challenges, x/(x - xi) and x/(x - w xi) at the query point;
    and the value the prover committed for the FRI polynomial at that point (first FRI tree)."""
    import json
    from tests import expr_real as R
    path = os.path.join(R.REF_DIR, f"{prog_name}.step52ns.xlpexpr")
    if not os.path.exists(path):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    F = json.load(open(os.path.join(golden_dir, proof_name)))
    code, ntmp1, ntmp3 = R.read_program(path)
    code, info = R.shrink(code, 1)                       # sections packed for a single row
    # where each section landed: shrink orders clusters by stride (row width)
    widths = sorted({len(F["queries"][0][f"s0_vals{t}"]) for t in ("1", "3", "4")})
    base, o = {}, 0
    for w in widths:
        base[w] = o; o += w
    ints = lambda v: [int(x) for x in v]
    nbits, nbits_ext = F["nBits"], F["nBitsExt"]
    xi = ints(F["challenges"]["7"])
    ch = np.zeros(8 * 3, np.uint64)
    for k, v in F["challenges"].items():
        ch[3 * int(k): 3 * int(k) + 3] = ints(v)
    evals = np.array([int(x) for e in F["evals"] for x in e], np.uint64)
    w_ext, w_n = orc.w(nbits_ext), orc.w(nbits)
    wxi = [orc.mul(c, w_n) for c in xi]
    cases = []
    for Q in F["queries"]:
        idx = Q["idx"]
        pols = np.zeros(info["pols_size"], np.uint64)
        for t in ("1", "3", "4"):
            vals = ints(Q[f"s0_vals{t}"])
            pols[base[len(vals)]: base[len(vals)] + len(vals)] = vals
        x = orc.mul(orc.shift(), orc.pow_(w_ext, idx))
        xd = orc.x_div_x_sub_xi(np.array([x], np.uint64), np.array(xi, np.uint64)).reshape(-1)
        xdw = orc.x_div_x_sub_xi(np.array([x], np.uint64), np.array(wxi, np.uint64)).reshape(-1)
        bufs = {"pols": pols, "const_pols_2ns": np.array(ints(Q["s0_valsC"]), np.uint64), "challenges": ch.copy(), "evals": evals.copy(),
                "x_div_x_sub_xi": xd.copy(), "x_div_x_sub_wxi": xdw.copy(), "f_2ns": np.zeros(3, np.uint64)}
        k = idx >> F["steps_nBits"][1]
        cases.append((bufs, ints(Q["s1_vals"])[3 * k: 3 * k + 3]))
    return code, ntmp1, ntmp3, len(F["queries"][0]["s0_valsC"]), cases


GOLDEN_FRI = GOLDEN_PROOFS


@pytest.mark.parametrize("proof,prog", GOLDEN_FRI)
def test_golden_proof_fri_polynomial_from_openings(orc, golden_dir, proof, prog):
    """GOLDEN-DATA PIN of the constraint-program path: the reference's step52ns program (converted), evaluated on the
    values a golden proof OPENS at each query, must reproduce the value of the FRI polynomial the reference prover
    committed there (first FRI tree of the same proof).  Ties together: the program conversion, operand semantics
    (sub13c / mul33c / accumulators), evals order, challenge numbering, xDivXSubXi and the FRI transposed layout."""
    code, ntmp1, ntmp3, nconst, cases = _golden_fri_polynomial_inputs(orc, golden_dir, proof, prog)
    assert len(cases) >= 2
    for bufs, expect in cases:
        orc.expr_run(code, ntmp1, ntmp3, bufs, 0, 1, 1, nconst)
        assert [int(v) for v in bufs["f_2ns"]] == expect


@pytest.mark.gpu
@pytest.mark.parametrize("proof,prog", GOLDEN_FRI)
def test_golden_proof_fri_polynomial_from_openings_gpu(xlp, orc, golden_dir, proof, prog):
    """the same golden check through the HIP VM"""
    code, ntmp1, ntmp3, nconst, cases = _golden_fri_polynomial_inputs(orc, golden_dir, proof, prog)
    p = xlp.ExprProgram(code, ntmp1, ntmp3)
    for bufs, expect in cases[:16]:
        dev = {k: xlp.from_np(v) for k, v in bufs.items()}
        p.run(xlp.steps_params(n_const_pols=1, n_const_pols_2ns=nconst, **dev), 0, 1); xlp.sync()
        assert [int(v) for v in xlp.to_np(dev["f_2ns"])] == expect
    p.close()


def _decode(code):
    """-> list of (op, (kind, dim, params) x 3) from the VM encoding"""
    from tests import expr_real as R
    out, cur = [], None
    for k, s, kind, dim, q in R.operands(code):
        prm = [int(v) for v in code[q:q + E.KIND_PARAMS[kind]]]
        if s == 0:
            cur = [None, (kind, dim, prm)]
            pc_head = q - 1
            cur[0] = int(code[pc_head]) & 0xff
        else:
            cur.append((kind, dim, prm))
        if s == 2:
            out.append(tuple(cur))
    return out


@pytest.mark.parametrize("proof,prog", GOLDEN_FRI)
def test_golden_proof_constraint_identity_at_xi(orc, golden_dir, proof, prog):
    """GOLDEN-DATA PIN of the quotient program: the verifier's identity C(xi) = Q(xi) Zh(xi) on a golden proof, with C
    computed by the reference's own step42ns program (converted) run symbolically at xi -- every column operand replaced
    by the proof's evaluation of that column (the evaluation map is read off the step52ns program: which column each
    evals[k] is subtracted from, and whether its group is multiplied by xDivXSubXi or xDivXSubWXi) -- and Q(xi)
    recombined from the evaluations of the cm4 columns: sum_p xi^(p N) q_p(xi)."""
    import json
    from tests import expr_real as R
    from tests.mini_verifier import add3, sub3, mul3, pow3, emb
    p42 = os.path.join(R.REF_DIR, f"{prog}.step42ns.xlpexpr"); p52 = os.path.join(R.REF_DIR, f"{prog}.step52ns.xlpexpr")
    if not (os.path.exists(p42) and os.path.exists(p52)):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    F = json.load(open(os.path.join(golden_dir, proof)))
    ints = lambda v: [int(x) for x in v]
    N = 1 << F["nBits"]
    evals = [ints(e) for e in F["evals"]]
    ch = {int(k): ints(v) for k, v in F["challenges"].items()}
    xi = ch[7]
    # ---- evaluation map from step52ns: evals[k] <-> (column, prime)
    code52, _, _ = R.read_program(p52)
    tags, evmap = {}, {}
    for op, d, a, b in _decode(code52):
        srcs = [a] + ([b] if b[0] != E.K_NONE else [])
        s = set()
        for kind, dim, prm in srcs:
            if kind == E.K_TMP3: s |= tags.get(prm[0], set())
        ev = [prm[0] for kind, dim, prm in srcs if kind == E.K_EVAL]
        if ev:
            col = [x for x in srcs if x[0] in (E.K_POLS, E.K_CONST2NS)]
            assert op == E.SUB and len(col) == 1
            key = ("pols", col[0][2][0]) if col[0][0] == E.K_POLS else ("const", col[0][2][0])
            evmap[ev[0]] = [key, None]
            s.add(ev[0])
        for kind, dim, prm in srcs:
            if kind in (E.K_XDIVXSUBXI, E.K_XDIVXSUBWXI):
                for k in s: evmap[k][1] = (kind == E.K_XDIVXSUBWXI)
                s = set()
        if d[0] == E.K_TMP3: tags[d[2][0]] = s
    assert len(evmap) == len(evals) and all(v[1] is not None for v in evmap.values())
    at = {(key, prime): evals[k] for k, (key, prime) in evmap.items()}
    # ---- step42ns at xi, every value in Fp^3
    code42, _, _ = R.read_program(p42)
    zh_inv = [int(v) for v in orc.inv3(sub3(pow3(xi, N), [1, 0, 0]))]
    t1, t3, Q = {}, {}, None

    def val(o):
        kind, dim, prm = o
        if kind == E.K_TMP1: return t1[prm[0]]
        if kind == E.K_TMP3: return t3[prm[0]]
        if kind == E.K_POLS: return at[(("pols", prm[0]), False)]
        if kind == E.K_POLS_PRIME: return at[(("pols", prm[0]), True)]
        if kind == E.K_CONST2NS: return at[(("const", prm[0]), False)]
        if kind == E.K_CONST2NS_PRIME: return at[(("const", prm[0]), True)]
        if kind == E.K_NUMBER: return emb(prm[0])
        if kind == E.K_CHALLENGE: return ch[prm[0]]
        if kind == E.K_PUBLIC: return emb(int(F["publics"][prm[0]]))
        if kind == E.K_X_2NS: return xi
        if kind == E.K_ZI: return zh_inv
        raise AssertionError(f"operand kind {kind} not expected in step42ns")
    for op, d, a, b in _decode(code42):
        x = val(a)
        r = x if op == E.COPY else add3(x, val(b)) if op == E.ADD else sub3(x, val(b)) if op == E.SUB else mul3(x, val(b))
        if d[0] == E.K_TMP1: t1[d[2][0]] = r
        elif d[0] == E.K_TMP3: t3[d[2][0]] = r
        elif d[0] == E.K_Q_2NS: Q = r
        else: raise AssertionError("unexpected destination")
    assert Q is not None
    # ---- Q(xi) from the cm4 evaluations: the widest-stride... cm4 is the section whose columns are all ext-3 quotient pieces
    strides = {}
    for op, d, a, b in _decode(code52):
        for kind, dim, prm in (a, b):
            if kind == E.K_POLS: strides.setdefault(prm[1], set()).add((prm[0], dim))
    cm4 = [s for s, cols in strides.items() if all(dim == 3 for _, dim in cols) and len(cols) * 3 == s]
    assert len(cm4) == 1
    pieces = sorted(off for off, _ in strides[cm4[0]])
    acc, xn = [0, 0, 0], [1, 0, 0]
    xiN = pow3(xi, N)
    for off in pieces:
        acc = add3(acc, mul3(xn, at[(("pols", off), False)])); xn = mul3(xn, xiN)
    assert Q == acc, "C(xi) / Zh(xi) computed by the reference's step42ns program != sum_p xi^(pN) q_p(xi) from the proof"


@pytest.mark.parametrize("src,fixture,prog", [(("finalProof/" if c == "recursive2" else "aggregatedProof/") + f[6:16] + ".zkin.proof_" + f[17:-5] + ".json", f, c)
                                              for f, c in GOLDEN_PROOFS])
def test_golden_proof_fri_polynomial_all_queries(orc, golden_dir, src, fixture, prog):
    """the same pin over EVERY query of the reference's proof files (the committed fixtures keep four each); runs only
    where the reference tree is mounted"""
    import json
    from tests import expr_real as R
    full = os.path.join("/root/reference/testvectors", src)
    path = os.path.join(R.REF_DIR, f"{prog}.step52ns.xlpexpr")
    if not (os.path.exists(full) and os.path.exists(path)):
        pytest.skip("reference tree / converted programs not present")
    J = json.load(open(full)); F = json.load(open(os.path.join(golden_dir, fixture)))
    ints = lambda v: [int(x) for x in v]
    code, ntmp1, ntmp3 = R.read_program(path)
    code, info = R.shrink(code, 1)
    widths = sorted({len(J[f"s0_vals{t}"][0]) for t in ("1", "3", "4")})
    base, o = {}, 0
    for w in widths:
        base[w] = o; o += w
    xi = ints(F["challenges"]["7"])
    ch = np.zeros(8 * 3, np.uint64)
    for k, v in F["challenges"].items():
        ch[3 * int(k): 3 * int(k) + 3] = ints(v)
    evals = np.array([int(x) for e in F["evals"] for x in e], np.uint64)
    w_ext, w_n = orc.w(F["nBitsExt"]), orc.w(F["nBits"])
    wxi = [orc.mul(c, w_n) for c in xi]
    nq = len(F["ys"])
    assert nq == len(J["s0_vals1"]) and nq >= 32
    for q, idx in enumerate(F["ys"]):
        pols = np.zeros(info["pols_size"], np.uint64)
        for t in ("1", "3", "4"):
            vals = ints(J[f"s0_vals{t}"][q])
            pols[base[len(vals)]: base[len(vals)] + len(vals)] = vals
        x = orc.mul(orc.shift(), orc.pow_(w_ext, idx))
        bufs = {"pols": pols, "const_pols_2ns": np.array(ints(J["s0_valsC"][q]), np.uint64), "challenges": ch.copy(), "evals": evals.copy(),
                "x_div_x_sub_xi": orc.x_div_x_sub_xi(np.array([x], np.uint64), np.array(xi, np.uint64)).reshape(-1).copy(),
                "x_div_x_sub_wxi": orc.x_div_x_sub_xi(np.array([x], np.uint64), np.array(wxi, np.uint64)).reshape(-1).copy(),
                "f_2ns": np.zeros(3, np.uint64)}
        orc.expr_run(code, ntmp1, ntmp3, bufs, 0, 1, 1, len(J["s0_valsC"][q]))
        k = idx >> F["steps_nBits"][1]
        assert [int(v) for v in bufs["f_2ns"]] == ints(J["s1_vals"][q])[3 * k: 3 * k + 3], q


@pytest.mark.gpu
def test_expr_row_windows_equal_full_domain(xlp, orc):
    """Row sharding (xlayer_prover_amd/shard.py): the HIP VM on two row WINDOWS of a domain (own rows + halo rows of the
    next shard, local row numbers, global modulus) reproduces the run over the whole domain.  Also: a window shorter than
    the highest next-row read is rejected."""
    import torch
    from xlayer_prover_amd import shard
    n, ninstr, halo = 4096, 200, 3
    prog, width = random_program(51, n, ninstr, prime_dst=False)
    full = make_buffers(orc, 0x5EED0D00, n, width, zi_size=4)
    widths = {"pols": width, "const_pols": 5, "const_pols_2ns": 4, "x_n": 1, "x_2ns": 1, "x_div_x_sub_xi": 3, "x_div_x_sub_wxi": 3, "q_2ns": 3, "f_2ns": 3}
    exp = {k: v.copy() for k, v in full.items()}
    orc.expr_run(prog.code, prog.ntmp1, prog.ntmp3, exp, 0, n, 5, 4)
    p = xlp.ExprProgram(prog.code, prog.ntmp1, prog.ntmp3)
    half = n // 2
    for s in range(2):
        r0 = s * half
        dev = {}
        for name, arr in full.items():
            if name in widths:
                w = widths[name]
                a2 = arr.reshape(n, w)
                win = np.concatenate([a2[r0:r0 + half], a2[(r0 + half) % n:(r0 + half) % n + halo]])     # what halo_exchange assembles
                dev[name] = xlp.from_np(np.ascontiguousarray(win).reshape(-1))
            else:
                dev[name] = xlp.from_np(arr)
        p.run(xlp.steps_params(n_const_pols=5, n_const_pols_2ns=4, **dev), 0, half); xlp.sync()
        for k in ("pols", "q_2ns", "f_2ns"):
            got = xlp.to_np(dev[k]).reshape(half + halo, widths[k])[:half]
            assert (got == exp[k].reshape(n, widths[k])[r0:r0 + half]).all(), (s, k)
        if s == 0:
            short = dict(dev); short["pols"] = dev["pols"][: (half + 0) * width]                            # no halo rows
            with pytest.raises(xlp.XlpError, match="past the end"):
                p.run(xlp.steps_params(n_const_pols=5, n_const_pols_2ns=4, **short), 0, half)
    p.close()
    # world-1 halo exchange = wrap-around copy
    t = torch.arange(12, dtype=torch.int64)
    assert shard.halo_exchange(t, 3, 1).tolist() == list(range(12)) + [0, 1, 2]


def test_expr_create_survives_malformed_programs():
    """xlp_expr_create is host-only (validation + compilation): mutated programs are either accepted or rejected with an
    error, never a crash; accepted ones also survive code generation"""
    import tempfile
    from xlayer_prover_amd import capi
    rng = np.random.default_rng(77)
    prog, _ = random_program(5, 64, 60)
    base = prog.code
    accepted = rejected = 0
    with tempfile.TemporaryDirectory() as d:
        for it in range(400):
            c = base.copy()
            for _ in range(int(rng.integers(1, 4))):
                k = int(rng.integers(0, c.size))
                mode = int(rng.integers(0, 3))
                c[k] = np.uint64(rng.integers(0, 1 << 63)) if mode == 0 else c[k] ^ np.uint64(1 << int(rng.integers(0, 40))) if mode == 1 else np.uint64(rng.integers(0, 64))
            if it % 7 == 0:
                c = c[: int(rng.integers(1, c.size))]
            try:
                p = capi.ExprProgram(c, prog.ntmp1, prog.ntmp3)
            except capi.XlpError:
                rejected += 1
                continue
            accepted += 1
            if accepted <= 20:
                p.codegen(os.path.join(d, "m.hip"))
            p.close()
    assert rejected > 50 and accepted > 10


def test_zkevm_geometry_relocation_is_consistent():
    """tools/zkevm_geometry.py: the packing rule reproduces the 11 section offsets recovered from the reference's step tables
    (SURVEY 8a9), and the five converted programs relocated to a 2^10-row slab address only their own sections"""
    from tests import expr_real as R
    from tools import zkevm_geometry as G
    G._check_real()
    if not os.path.exists(os.path.join(R.REF_DIR, "zkevm.step42ns.xlpexpr")):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    progs, info = G.load(10)
    L = G.layout(10)
    assert info["nev"] == 1973 and info["npub"] == 44 and max(info["n_const"], info["n_const2"]) == 234
    for step, (code, _, _) in progs.items():
        n_dom = 1 << (11 if step.endswith("2ns") else 10)
        for _, _, kind, dim, q in R.operands(code):
            if kind in (E.K_POLS, E.K_POLS_PRIME):
                off = int(code[q]); stride = int(code[q + (1 if kind == E.K_POLS else 3)])
                name = [s for s, _, _ in G.SECTIONS if L[s][0] <= off][-1]
                assert stride == L[name][1] and off - L[name][0] + dim <= stride, (step, name)
                assert L[name][2] >= n_dom                       # a base-domain step never reads an extended section and vice versa
            if kind in (E.K_POLS_PRIME, E.K_CONST_PRIME, E.K_CONST2NS_PRIME):
                assert int(code[q + 2]) == n_dom, step


# ------------------------------------------------------------------------------- Horner chains -> dot products (expr_opt.py)
OPT_PROGS = [f"{stark}.{step}" for stark in ("zkevm", "recursive1", "recursive2", "c12a", "recursivef")
             for step in ("step2prev", "step3prev", "step3", "step42ns", "step52ns") if not (step == "step2prev" and stark != "zkevm")]


def _opt_case(orc, name, n):
    """-> (original code + temporaries, optimised row program, table program, n_ext, nch, buffers, info, stats) for a converted program"""
    from tests import expr_real as R
    from xlayer_prover_amd import expr_opt
    path = os.path.join(R.REF_DIR, f"{name}.xlpexpr")
    if not os.path.exists(path):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    code, n1, n3 = R.read_program(path)
    nch = 8
    row, tab, n_ext, stats = expr_opt.horner_to_dot(code, n1, n3, nch)
    (c0, c1), info = R.shrink_many([code, row.code], n)
    bufs = R.buffers(lambda cnt, seed: orc.fill(cnt, seed), 11, n, info)
    bufs["challenges"] = orc.fill(3 * nch, 99)
    return (c0, n1, n3), (c1, row.ntmp1, row.ntmp3), tab, n_ext, nch, bufs, info, stats


@pytest.mark.parametrize("name", OPT_PROGS)
def test_horner_to_dot_equals_original_on_the_oracle(orc, name):
    """the rewritten step programs (powers of the challenges and the evaluation constants from a one-row table program,
    base-field columns multiplied by extension constants) give the values of the reference's programs, bit for bit"""
    n = 32
    (c0, n1, n3), (c1, m1, m3), tab, n_ext, nch, b0, info, stats = _opt_case(orc, name, n)
    if name.endswith("step52ns"):
        assert stats["ext_mul_after"] * 4 < stats["ext_mul_before"]
    b1 = {k: v.copy() for k, v in b0.items()}
    orc.expr_run(c0, n1, n3, b0, 0, n, info["n_const"], info["n_const2"])
    ext = np.zeros(3 * n_ext, np.uint64); ext[:3 * nch] = b1["challenges"]
    if tab.ninstr:
        orc.expr_run(tab.code, tab.ntmp1, tab.ntmp3, {**b1, "pols": ext}, 0, 1)
    b1["challenges"] = ext
    orc.expr_run(c1, m1, m3, b1, 0, n, info["n_const"], info["n_const2"])
    for k in ("pols", "q_2ns", "f_2ns"):
        assert (b0[k] == b1[k]).all(), k


def test_horner_to_dot_removes_most_extension_products_of_the_zkevm_steps(orc):
    from tests import expr_real as R
    from xlayer_prover_amd import expr_opt
    if not os.path.exists(os.path.join(R.REF_DIR, "zkevm.step42ns.xlpexpr")):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    before = after = 0
    for step in ("step2prev", "step3prev", "step3", "step42ns", "step52ns"):
        code, n1, n3 = R.read_program(os.path.join(R.REF_DIR, f"zkevm.{step}.xlpexpr"))
        st = expr_opt.horner_to_dot(code, n1, n3, 8)[3]
        before += st["ext_mul_before"]; after += st["ext_mul_after"]
    assert before == 8243 and after < 1500


@pytest.mark.gpu
@pytest.mark.parametrize("name,n", [("zkevm.step52ns", 1 << 12), ("zkevm.step42ns", 1 << 10), ("zkevm.step3prev", 3000), ("recursive2.step42ns", 1000)])
def test_horner_to_dot_on_the_gpu_equals_original_on_the_oracle(xlp, orc, name, n):
    (c0, n1, n3), (c1, m1, m3), tab, n_ext, nch, b0, info, _ = _opt_case(orc, name, n)
    dev = {k: xlp.from_np(v) for k, v in b0.items()}
    ext = xlp.from_np(np.concatenate([b0["challenges"], np.zeros(3 * (n_ext - nch), np.uint64)]))
    if tab.ninstr:
        pt = xlp.ExprProgram(tab.code, tab.ntmp1, tab.ntmp3)
        pt.run(xlp.steps_params(pols=ext, challenges=dev["challenges"], evals=dev["evals"]), 0, 1)
        pt.close()
    pr = xlp.ExprProgram(c1, m1, m3)
    pr.run(xlp.steps_params(n_const_pols=info["n_const"], n_const_pols_2ns=info["n_const2"], **{**dev, "challenges": ext}), 0, n)
    xlp.sync()
    orc.expr_run(c0, n1, n3, b0, 0, n, info["n_const"], info["n_const2"])
    for k in ("pols", "q_2ns", "f_2ns"):
        assert (xlp.to_np(dev[k]) == b0[k]).all(), k
    pr.close()


def _chainy_program(seed, n, ninstr, nin=10, nstore=30):
    """random programs made of the shapes horner_to_dot reasons about -- shifts by a challenge, sums / differences of
    columns, `col - evals[e]`, temporaries that are copies of columns or computed, constants -- mixed with the things that
    must stop it: extension products of temporaries, stores to columns that are read back later, recycled temporaries"""
    rng = np.random.default_rng(seed)
    width = nin + 3 * nstore
    p = E.Program()
    written = []                                             # (first column, dim) stored so far: readable on the same row
    nxt = [nin]
    t3 = lambda: E.tmp3(int(rng.integers(0, 3)))
    t1 = lambda: E.tmp1(int(rng.integers(0, 4)))
    ch = lambda: E.challenge(int(rng.integers(0, 3)))

    def col(dim=None):
        dim = dim or (1 if rng.integers(0, 3) else 3)
        k = int(rng.integers(0, 6))
        if k == 0 and written:
            c, d = written[int(rng.integers(0, len(written)))]
            if d >= dim: return E.pols(c, width, dim)
        if k == 1: return E.pols_prime(int(rng.integers(0, nin - 2)), 1, n, width, dim)
        if k == 2 and dim == 1: return E.const2ns(int(rng.integers(0, 4)))
        if k == 3 and dim == 1: return E.const(int(rng.integers(0, 5)))
        return E.pols(int(rng.integers(0, nin - 2)), width, dim)

    def term():
        k = int(rng.integers(0, 8))
        if k <= 2: return col()
        if k == 3: return t1()
        if k == 4: return E.number(int(rng.integers(0, 1 << 62)))
        return t3()
    # every temporary starts defined
    for i in range(4): p.copy(E.tmp1(i), col(1))
    for i in range(3): p.mul(E.tmp3(i), col(), ch())
    for _ in range(ninstr):
        k = int(rng.integers(0, 14))
        if k <= 2: p.mul(t3(), term(), ch())
        elif k == 3: p.mul(t3(), ch(), term())
        elif k <= 6:
            a, b = t3(), term()
            if rng.integers(0, 2): a, b = b, a
            (p.add if rng.integers(0, 3) else p.sub)(t3(), a, b)
        elif k == 7: p.sub(t3(), col(), E.evaluation(int(rng.integers(0, 5))))
        elif k == 8: p.copy(t1(), col(1))
        elif k == 9: p.mul(t1(), col(1), t1())
        elif k == 10: p.mul(t3(), t3(), t3())
        elif k == 11 and nxt[0] + 3 <= width:
            d = 1 if rng.integers(0, 2) else 3
            if d == 3: p.add(E.pols(nxt[0], width, 3), t3(), term())
            else: p.add(E.pols(nxt[0], width, 1), t1(), col(1))
            written.append((nxt[0], d)); nxt[0] += 3
        elif k == 12: p.copy(E.f_2ns() if rng.integers(0, 2) else E.q_2ns(), t3())
        else: p.mul(t3(), t3(), E.x_div_x_sub_xi())
    for i in range(3):
        if nxt[0] + 3 <= width:
            p.copy(E.pols(nxt[0], width, 3), E.tmp3(i)); nxt[0] += 3
    return p, width


@pytest.mark.parametrize("block", range(8))
def test_horner_to_dot_fuzz_on_the_oracle(orc, block):
    """400 random chain-shaped programs: the rewritten program + its table program give the original's outputs"""
    from xlayer_prover_amd import expr_opt
    n, nch, removed = 6, 4, 0
    for seed in range(50 * block, 50 * block + 50):
        prog, width = _chainy_program(seed, n, 30 + seed % 40)
        row, tab, n_ext, st = expr_opt.horner_to_dot(prog.code, prog.ntmp1, prog.ntmp3, nch)
        removed += st["ext_mul_before"] - st["ext_mul_after"]
        b0 = make_buffers(orc, seed, n, width, nch=nch)
        b0["pols"] = orc.fill(n * width, seed + 77)          # canonical inputs, as the prover's sections are
        b1 = {k: v.copy() for k, v in b0.items()}
        orc.expr_run(prog.code, prog.ntmp1, prog.ntmp3, b0, 0, n, 5, 4)
        ext = np.zeros(3 * n_ext, np.uint64); ext[:3 * nch] = b1["challenges"]
        if tab.ninstr:
            orc.expr_run(tab.code, tab.ntmp1, tab.ntmp3, {**b1, "pols": ext}, 0, 1, 5, 4)
        b1["challenges"] = ext
        orc.expr_run(row.code, row.ntmp1, row.ntmp3, b1, 0, n, 5, 4)
        for k in ("pols", "q_2ns", "f_2ns"):
            assert (b0[k] == b1[k]).all(), (seed, k)
    assert removed > 0


@pytest.mark.parametrize("seed", range(12))
def test_hoist_sum_terms_fuzz_on_the_oracle(orc, seed):
    """expr_opt.hoist_sum_terms on programs shaped like the optimiser's output: temporaries produced early (some overwritten
    again, some shared by two sums, some extension-valued), unrelated work in between, then runs of `mul S, x, K; add|sub D, D, S`.
    The hoisted program gives the same outputs, and the pairs really moved."""
    from xlayer_prover_amd import expr_opt
    rng = np.random.default_rng(1000 + seed)
    n, nch, width, nt1, nt3 = 5, 6, 40, 24, 6
    p = E.Program()
    col = lambda d=1: E.pols(int(rng.integers(0, 30)), width, d)
    ch = lambda: E.challenge(int(rng.integers(0, nch)))
    for i in range(nt1):
        p.mul(E.tmp1(i), col(), col()) if rng.integers(0, 2) else p.add(E.tmp1(i), col(), E.number(int(rng.integers(1, 99))))
    for i in range(nt3):
        p.mul(E.tmp3(i), col(3), ch())
    for _ in range(int(rng.integers(5, 40))):                  # filler: rewrites some temporaries after their first definition
        k = int(rng.integers(0, 4))
        if k == 0: p.mul(E.tmp1(int(rng.integers(0, nt1))), col(), col())
        elif k == 1: p.add(E.tmp3(int(rng.integers(2, nt3))), E.tmp3(int(rng.integers(2, nt3))), col(3))
        else: p.sub(E.tmp1(int(rng.integers(0, nt1))), E.tmp1(int(rng.integers(0, nt1))), col())
    S = E.tmp3(nt3)
    for D in (E.tmp3(0), E.tmp3(1)):                           # two sums; D starts from a real value
        for _ in range(int(rng.integers(3, 30))):
            x = E.tmp1(int(rng.integers(0, nt1))) if rng.integers(0, 4) else E.tmp3(int(rng.integers(2, nt3)))
            if rng.integers(0, 2): p.mul(S, x, ch())
            else: p.mul(S, ch(), x)
            (p.add if rng.integers(0, 3) else p.sub)(D, D, S)
        p.mul(D, D, ch())                                      # the sum is consumed by something that is not part of a run
    p.copy(E.q_2ns(), E.tmp3(0)); p.copy(E.f_2ns(), E.tmp3(1)); p.copy(E.pols(33, width, 3), E.tmp3(2))
    q = expr_opt.hoist_sum_terms(p, min_distance=1, min_terms=1)
    assert q.ninstr > p.ninstr                                 # zeroing + the `add D, D, A` that stays
    b0 = make_buffers(orc, seed, n, width, nch=nch)
    b0["pols"] = orc.fill(n * width, seed + 5)
    b1 = {k: v.copy() for k, v in b0.items()}
    orc.expr_run(p.code, p.ntmp1, p.ntmp3, b0, 0, n, 5, 4)
    orc.expr_run(q.code, q.ntmp1, q.ntmp3, b1, 0, n, 5, 4)
    for k in ("pols", "q_2ns", "f_2ns"):
        assert (b0[k] == b1[k]).all(), k
    assert expr_opt.hoist_sum_terms(p, min_distance=10 ** 6) is p          # nothing far enough: the program comes back as it is


def _dots_on_host(orc, dots, bufs, ext, n):
    """xlp_dev_dot_columns restated with Python integers on the buffers of tests/expr_real.py: -> uint64 [n][3 * ngroups]"""
    Pm = 0xFFFFFFFF00000001
    G = dots["ngroups"]
    out = np.zeros((n, 3 * G), np.uint64)
    nconst2 = bufs["const_pols_2ns"].size // n
    for row in range(n):
        acc = [[0, 0, 0] for _ in range(G)]
        for x, g, k in dots["entries"]:
            if x.kind == E.K_POLS:
                v = bufs["pols"][x.params[0] + row * x.params[1]: x.params[0] + row * x.params[1] + x.dim]
            else:
                assert x.kind == E.K_CONST2NS
                v = bufs["const_pols_2ns"][x.params[0] + row * nconst2: x.params[0] + row * nconst2 + 1]
            v = [int(t) for t in v] + [0] * (3 - len(v))
            m = orc.mul3(np.array([int(t) for t in ext[3 * k:3 * k + 3]], np.uint64), np.array(v, np.uint64))
            acc[g] = [(a + int(b)) % Pm for a, b in zip(acc[g], m)]
        out[row] = [c for a in acc for c in a]
    return out


@pytest.mark.parametrize("prog", ["zkevm", "recursive1", "recursive2", "c12a", "recursivef"])
def test_fri_polynomial_as_dot_columns_equals_original_on_the_oracle(orc, prog):
    """step52ns with its three big sums handed to the dot-columns kernel (restated on the host here) and a 12-instruction
    row program on the kernel's output: same f_2ns as the reference's program"""
    from tests import expr_real as R
    from xlayer_prover_amd import expr_opt
    path = os.path.join(R.REF_DIR, f"{prog}.step52ns.xlpexpr")
    if not os.path.exists(path):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    code, n1, n3 = R.read_program(path)
    n, nch = 32, 8
    (c0,), info = R.shrink_many([code], n)
    row, tab, n_ext, st, dots = expr_opt.horner_to_dot(c0, n1, n3, nch, outsource_min_terms=8)
    if dots is None:            # a sum with a non-column term keeps the program reading real columns: the inline form is returned
        assert prog in ("c12a", "recursivef") and row.ninstr > 100
        return
    assert row.ninstr < 40 and st["ext_mul_after"] <= 4
    b0 = R.buffers(lambda cnt, seed: orc.fill(cnt, seed), 11, n, info)
    b0["challenges"] = orc.fill(3 * nch, 99)
    b1 = {k: v.copy() for k, v in b0.items()}
    orc.expr_run(c0, n1, n3, b0, 0, n, info["n_const"], info["n_const2"])
    ext = np.zeros(3 * n_ext, np.uint64); ext[:3 * nch] = b1["challenges"]
    orc.expr_run(tab.code, tab.ntmp1, tab.ntmp3, {**b1, "pols": ext}, 0, 1)
    dots_out = _dots_on_host(orc, dots, b1, ext, n)
    b2 = {**b1, "pols": dots_out.reshape(-1).copy(), "challenges": ext}
    orc.expr_run(row.code, row.ntmp1, row.ntmp3, b2, 0, n, info["n_const"], info["n_const2"])
    assert (b2["f_2ns"] == b0["f_2ns"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("prog,n", [("zkevm", 3000), ("recursive2", 1 << 12)])
def test_fri_polynomial_as_dot_columns_on_the_gpu(xlp, orc, prog, n):
    """table program (VM, one row) -> xlp_dev_dot_columns over the sections -> 12-instruction row program (VM), all on the GPU,
    against the reference's step52ns on the oracle"""
    import torch
    from tests import expr_real as R
    from xlayer_prover_amd import expr_opt
    path = os.path.join(R.REF_DIR, f"{prog}.step52ns.xlpexpr")
    if not os.path.exists(path):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    code, n1, n3 = R.read_program(path)
    nch = 8
    (c0,), info = R.shrink_many([code], n)
    row, tab, n_ext, st, dots = expr_opt.horner_to_dot(c0, n1, n3, nch, outsource_min_terms=8)
    b0 = R.buffers(lambda cnt, seed: orc.fill(cnt, seed), 11, n, info)
    b0["challenges"] = orc.fill(3 * nch, 99)
    dev = {k: xlp.from_np(v) for k, v in b0.items()}
    ext = xlp.from_np(np.concatenate([b0["challenges"], np.zeros(3 * (n_ext - nch), np.uint64)]))
    pt = xlp.ExprProgram(tab.code, tab.ntmp1, tab.ntmp3)
    pt.run(xlp.steps_params(pols=ext, challenges=dev["challenges"], evals=dev["evals"]), 0, 1)
    G = dots["ngroups"]
    nconst2 = info["n_const2"]
    ents = [(dev["pols"], x.params[1], x.params[0], x.dim, g, k) if x.kind == E.K_POLS else (dev["const_pols_2ns"], nconst2, x.params[0], 1, g, k)
            for x, g, k in dots["entries"]]
    dots_out = torch.zeros(n * 3 * G, dtype=torch.int64, device="cuda")
    xlp.dev_dot_columns(dots_out, ents, G, ext, n)
    pr = xlp.ExprProgram(row.code, row.ntmp1, row.ntmp3)
    pr.run(xlp.steps_params(n_const_pols=info["n_const"], n_const_pols_2ns=nconst2, **{**dev, "pols": dots_out, "challenges": ext}), 0, n)
    xlp.sync()
    orc.expr_run(c0, n1, n3, b0, 0, n, info["n_const"], info["n_const2"])
    assert (xlp.to_np(dev["f_2ns"]) == b0["f_2ns"]).all()
    pt.close(); pr.close()
