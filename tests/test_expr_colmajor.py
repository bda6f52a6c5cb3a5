"""Column-major step programs (include/xlp_gl.h, XLP_EXPR_COLMAJOR; tools/zkevm_geometry.py to_colmajor): the reference's converted constraint
programs, relocated to one consistent slab, run (a) on the oracle in the reference's row-major addressing and (b) on the HIP VM -- interpreter and
ahead-of-time kernels -- in the column-major / coset-major addressing on the transposed buffers.  Same field elements, cell for cell."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def xlp():
    import torch
    from xlayer_prover_amd import capi
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.init(0)
    capi.use_torch_stream()
    return capi


def _programs(circ, ln):
    from tools import zkevm_geometry as G
    from xlayer_prover_amd import programs as R
    if not os.path.exists(os.path.join(R.REF_DIR, f"{circ}.step52ns.xlpexpr")):
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    progs, info = G.load(ln, False, circ)
    return G, progs, info, G.layout(ln, False, circ), G.CIRCUITS[circ]["ext"]


def _buffers(orc, L, info, n, ext, seed):
    ne = n << ext
    nc = max(info["n_const"], info["n_const2"])
    fill = lambda k, words: orc.fill(words, seed + k)
    return {"pols": fill(0, L["end"][0]), "const_pols": fill(1, n * nc), "const_pols_2ns": fill(2, ne * nc), "challenges": fill(3, 3 * max(8, info["nch"])),
            "public_inputs": fill(4, info["npub"]), "evals": fill(5, 3 * info["nev"]), "x_n": fill(6, n), "x_2ns": fill(7, ne), "zi": fill(8, 1 << ext),
            "x_div_x_sub_xi": fill(9, 3 * ne), "x_div_x_sub_wxi": fill(10, 3 * ne), "q_2ns": fill(11, 3 * ne), "f_2ns": fill(12, 3 * ne)}, nc


def _to_colmajor_buffers(G, bufs, L, n, ext, nc):
    ne = n << ext
    cm = dict(bufs)
    cm["pols"] = G.slab_to_colmajor(bufs["pols"], L, ext)
    cm["const_pols"] = G.matrix_to_colmajor(bufs["const_pols"].reshape(n, nc), 0)
    cm["const_pols_2ns"] = G.matrix_to_colmajor(bufs["const_pols_2ns"].reshape(ne, nc), ext)
    cm["x_2ns"] = G.rows_to_storage(bufs["x_2ns"].reshape(ne, 1), ext).reshape(-1)
    for k in ("x_div_x_sub_xi", "x_div_x_sub_wxi"):
        cm[k] = G.rows_to_storage(bufs[k].reshape(ne, 3), ext).reshape(-1)
    return {k: np.ascontiguousarray(v) for k, v in cm.items()}      # q_2ns / f_2ns / zi / x_n / challenges ...: as they are


# (the small STARKs run at their real geometry, 2^17+ rows: their column-major programs are covered by the C++ proofs of tests/test_cpp_colmajor.py)
CASES = [("zkevm", "step2prev"), ("zkevm", "step3prev"), ("zkevm", "step3"), ("zkevm", "step42ns"), ("zkevm", "step52ns")]


@pytest.mark.gpu
@pytest.mark.parametrize("circ,step", CASES)
def test_colmajor_program_on_the_gpu_equals_the_rowmajor_program_on_the_oracle(xlp, orc, tmp_path, monkeypatch, circ, step):
    ln = 8
    G, progs, info, L, ext = _programs(circ, ln)
    code, n1, n3 = progs[step]
    n = 1 << ln
    rows = n << ext if step.endswith("2ns") else n
    bufs, nc = _buffers(orc, L, info, n, ext, 0x5EED0C00)
    exp = {k: v.copy() for k, v in bufs.items()}
    orc.expr_run(code, n1, n3, exp, 0, rows, nc, nc)
    assert any((exp[k] != bufs[k]).any() for k in ("pols", "q_2ns", "f_2ns"))
    code_cm = G.to_colmajor(code, L, ext)
    cm = _to_colmajor_buffers(G, bufs, L, n, ext, nc)
    # the ahead-of-time kernels are generated the way the product's plugins are (chelpers/Makefile GEN: loads pipelined, segment cuts by dynamic
    # programming) and under the promise of canonical sections (XLP_EXPR_CANONICAL: the buffers here are orc.fill's, canonical)
    p = xlp.ExprProgram(code_cm, n1, n3, colmajor=True, canonical=True)
    # (step52ns: the product runs the FRI polynomial on xlp_dev_dot_columns_cm, not as generated kernels -- interpreter only here, its build is the longest)
    for compiled in ((False,) if step == "step52ns" else (False, True)):
        if compiled:
            # the plugin `make -C xlayer_prover_amd/chelpers colmajor-plugins` built for this geometry (attach refuses it unless it was generated from exactly this
            # program under these flags); built here only if it is not there
            from xlayer_prover_amd import programs as R
            so = os.path.join(R.REF_DIR, f"{circ}.{step}.geo{ln}c.so")
            if os.path.exists(so):
                p.attach(so)
            else:
                monkeypatch.setenv("XLP_EXPR_HOIST_LOADS", "4"); monkeypatch.setenv("XLP_EXPR_CUT", "320")
                p.compile(str(tmp_path / "jit"), parts=16)
        dev = {k: xlp.from_np(v) for k, v in cm.items()}
        sp = xlp.steps_params(n_const_pols=nc, n_const_pols_2ns=nc, coset_rows=n if step.endswith("2ns") else 0, **dev)
        p.run(sp, 0, rows); xlp.sync()
        got = G.slab_from_colmajor(xlp.to_np(dev["pols"]), L, ext)
        bad = np.flatnonzero(got != exp["pols"])
        assert bad.size == 0, (compiled, bad[:8])
        for k in ("q_2ns", "f_2ns"):
            assert (xlp.to_np(dev[k]) == exp[k]).all(), (compiled, k)
    p.close()


@pytest.mark.gpu
def test_colmajor_program_on_a_row_range_and_in_several_slabs(xlp, orc, tmp_path, monkeypatch):
    """a dense coset-major row range that starts and ends inside cosets, in slabs of 256 rows on two streams (XLP_EXPR_SLAB_WGS): only the
    rows of the range change"""
    G, progs, info, L, ext = _programs("zkevm", 10)
    code, n1, n3 = progs["step42ns"]
    n = 1 << 10
    bufs, nc = _buffers(orc, L, info, n, ext, 0x5EED0C40)
    cm = _to_colmajor_buffers(G, bufs, L, n, ext, nc)
    p = xlp.ExprProgram(G.to_colmajor(code, L, ext), n1, n3, colmajor=True)
    d0, cnt = 700, 900                                               # dense indices: coset 0 rows 700.., coset 1 rows ..575
    dev = {k: xlp.from_np(v) for k, v in cm.items()}
    sp = xlp.steps_params(n_const_pols=nc, n_const_pols_2ns=nc, coset_rows=n, **dev)
    p.run(sp, d0, cnt); xlp.sync()
    p.close()
    # the same natural rows on the oracle, one by one (its ranges are natural rows)
    nat = [((d % n) << ext) | (d // n) for d in range(d0, d0 + cnt)]
    exp = {k: v.copy() for k, v in bufs.items()}
    for k0 in sorted(nat):
        orc.expr_run(code, n1, n3, exp, k0, 1, nc, nc)
    assert (xlp.to_np(dev["q_2ns"]) == exp["q_2ns"]).all()
    assert (G.slab_from_colmajor(xlp.to_np(dev["pols"]), L, ext) == exp["pols"]).all()


@pytest.mark.parametrize("world", [0, 2])
def test_to_colmajor_addresses_the_same_cells(world):
    """no GPU: every operand of the converted program names, for every row, the cell its row-major original names (checked through the
    transposition of a slab that holds its own word indices).  world: a rank's WINDOW of a row-sharded proof (its rows + the halo row; an
    extended column = 2^ext coset blocks of rows + halo row): a next-row operand of the last row names the halo word of its coset block"""
    from tools import zkevm_geometry as G
    from xlayer_prover_amd import expr as E
    from xlayer_prover_amd import programs as R
    if not os.path.exists(os.path.join(R.REF_DIR, "zkevm.step52ns.xlpexpr")):
        pytest.skip("converted reference programs not built")
    ln, ext = 5, 1
    L = G.layout_window(ln, world) if world else G.layout(ln)
    progs, info = G.load(ln, False, "zkevm", L if world else None)
    n = (1 << ln) // (world or 1)                                       # rows of the program's range (a rank's own rows)
    ident = np.arange(L["end"][0], dtype=np.uint64)
    cmslab = G.slab_to_colmajor(ident, L, ext)                          # cmslab[storage address] = row-major address of the cell stored there
    for step, (code, _, _) in progs.items():
        rows = n << ext if step.endswith("2ns") else n
        cm = G.to_colmajor(code, L, ext)
        nb = (n + 1 if world else n) if step.endswith("2ns") else rows  # storage rows of a coset block
        for (op, d, a, b), (_, dc, ac, bc) in zip(E.decode(code), E.decode(cm)):
            for o, oc in ((d, dc), (a, ac), (b, bc)):
                if o is None or o.kind not in (E.K_POLS, E.K_POLS_PRIME):
                    continue
                for k in (0, 1, rows // 2 + 1, rows - 1):               # natural row k
                    r, q = (k & ((1 << ext) - 1), k >> ext) if step.endswith("2ns") else (0, k)
                    s_ = r * nb + q                                      # its storage row
                    for c in range(o.dim):
                        if o.kind == E.K_POLS:
                            want = o.params[0] + k * o.params[1] + c
                            got = cmslab[oc.params[0] + c * oc.params[1] + s_]
                        else:
                            off, nx, md, st = o.params
                            want = off + ((k + nx) % md) * st + c
                            offc, nxc, mdc, cs = oc.params
                            j = s_ + nxc
                            if j - r * nb >= mdc:
                                j -= mdc
                            got = cmslab[offc + c * cs + j]
                        assert int(got) == want, (step, k, c)
