#!/usr/bin/env python3
"""Ahead-of-time plugins of the converted step programs (xlp_expr_codegen -> hipcc -> <name>.so next to the programs in xlayer_prover_amd/chelpers: git-ignored,
travels to the GPU box).  Needs hipcc, no GPU.  Two families:
    --log-rows 8,19                         the zkEVM programs on the synthetic slab of tests/expr_real.py              -> zkevm.<step>.n<log2 n>.so
    --geometry N [--circuit C] [--aliased] [--world G] [--colmajor] [--opt-only | --plain-only]
                                            relocated to a zkEVM-geometry slab / a rank's window (tools/zkevm_geometry.py) -> <C>.<step>[_opt].geo<N>[a][wG][c].so
Generator options (read when the text is written; the stamp covers the text, so a change rebuilds): --hoist-loads, --cut / --seg, --tune STEP=HOIST,CUT per program.
After each build: the generator's account of what the kernels move per row, and what the COMPILED code does (tools/isa_loads_in_flight.py), printed and written to
<name>.report.json.  xlayer_prover_amd/chelpers/Makefile holds the product's settings (GEN, TUNE)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import expr_real as R
from xlayer_prover_amd import capi


def headers_hash():
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(os.path.dirname(os.path.abspath(capi.__file__)), "csrc")
    for f in ("gl_field.hpp", "xlp_expr_jit.h"):
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-rows", default="8,19")
    ap.add_argument("--steps", default="step2prev,step3prev,step3,step42ns,step52ns")
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--full", action="store_true", help="full zkEVM domains instead of --log-rows: 2^23 rows for the *prev/step3 programs, 2^24 for the 2ns ones")
    ap.add_argument("--geometry", type=int, default=0, help="log2 N: the five programs relocated to ONE zkEVM-geometry slab of 2^N rows "
                    "(tools/zkevm_geometry.py, used by tools/bench_zkevm_genproof.py) -> zkevm.<step>.geo<N>.so")
    ap.add_argument("--circuit", default="zkevm", help="with --geometry: zkevm, c12a or recursive1 (tools/zkevm_geometry.py CIRCUITS; the small ones only at their real size)")
    ap.add_argument("--opt-only", action="store_true", help="with --geometry: only the dot-product forms of steps 2-4 (what the genProof bench runs by default; the one-row table "
                    "program of step52ns costs the same few ms interpreted and the longest build)")
    ap.add_argument("--drow", action="store_true", help="with --geometry: also the 12-instruction row program of the dot-columns form of step52ns (-> <circuit>.step52ns_drow.geo<N>.so; "
                    "measured in round 5: nothing gained over the interpreter, so the bundles ship without it)")
    ap.add_argument("--plain-only", action="store_true", help="with --geometry: only the programs as converted (no dot-product forms, no table / row programs): what "
                    "tests/test_expr_colmajor.py attaches at 2^8 rows instead of compiling on the GPU box")
    ap.add_argument("--aliased", action="store_true", help="with --geometry: the one-GPU aliased slab layout -> zkevm.<step>.geo<N>a.so")
    ap.add_argument("--world", type=int, default=0, help="with --geometry: the window layout of one rank of a proof row-sharded over this many ranks "
                    "(tools/zkevm_geometry.py layout_window) -> zkevm.<step>.geo<N>w<G>.so")
    ap.add_argument("--colmajor", action="store_true", help="with --geometry: column-major device sections (tools/zkevm_geometry.py to_colmajor; XLP_EXPR_COLMAJOR) "
                    "-> zkevm.<step>.geo<N>[a]c.so")
    ap.add_argument("--hoist-loads", type=int, default=0, help="the generator pipelines the loads of a segment in batches of this many "
                    "(XLP_EXPR_HOIST_LOADS, csrc/expr.hip place_loads; 0 = every load at its first use)")
    ap.add_argument("--seg", type=int, default=0, help="a segment boundary every this many instructions (XLP_EXPR_SEG; default: the generator's 256)")
    ap.add_argument("--cut", type=int, default=0, help="segment boundaries by dynamic programming, segments of at most this many instructions (XLP_EXPR_CUT)")
    ap.add_argument("--tune", action="append", default=[], metavar="STEP=HOIST,CUT", help="per-program generator options that override --hoist-loads / --cut for that step "
                    "(e.g. --tune step3prev=6,256 --tune step42ns=4,512: the programs differ in what they like -- profiles/r5/step_program_variant_sweep.txt)")
    ap.add_argument("--out-dir", default="", help="write the plugins (and their build directory) here instead of xlayer_prover_amd/chelpers (experiments)")
    a = ap.parse_args()
    # the generator reads its options when it writes the text: the stamp below covers the text, so a change of an option rebuilds the plugin
    for k, v in (("XLP_EXPR_HOIST_LOADS", a.hoist_loads), ("XLP_EXPR_SEG", a.seg), ("XLP_EXPR_CUT", a.cut)):
        if v:
            os.environ[k] = str(v)
    sizes = ["full"] if a.full else [int(x) for x in a.log_rows.split(",")]
    geo = None
    if a.geometry:
        from tools import zkevm_geometry as G
        if a.steps == ap.get_default("steps"):
            a.steps = ",".join(G.CIRCUITS[a.circuit]["steps"])
        LW = G.layout_window(a.geometry, a.world, a.circuit) if a.world else None
        geo, _ = G.load(a.geometry, a.aliased, a.circuit, LW)
        for st, v in G.load_opt(a.geometry, a.aliased, a.circuit, LW).items():
            geo[st + "_opt"] = v[0]
        sizes = ["geo"]
        want52 = "step52ns" in a.steps.split(",")
        a.steps = ",".join(x for st in a.steps.split(",") for x in ((st + "_opt",) if a.opt_only else (st,) if a.plain_only else (st, st + "_opt")) if x != "step52ns_opt" or not a.opt_only)
        if a.plain_only:
            want52 = False
        if want52 and not a.opt_only:                     # the one-row table program of the dot-columns form of step52ns
            tabp = G.load_fri_dots(a.geometry, a.aliased, a.circuit, LW)[0]
            geo["step52ns_tab"] = (tabp.code, tabp.ntmp1, tabp.ntmp3)
            a.steps += ",step52ns_tab"
        drow_layout = None
        if want52 and a.drow:                             # ... and its 12-instruction ROW program (reads the sums xlp_dev_dot_columns[_cm] leaves: 3 x ngroups columns;
            _, dots, rowp, _ = G.load_fri_dots(a.geometry, a.aliased, a.circuit, LW)     # interpreted it cost 27 ms of a zkEVM proof: round 5)
            geo["step52ns_drow"] = (rowp.code, rowp.ntmp1, rowp.ntmp3)
            a.steps += ",step52ns_drow"
            Ld = LW or G.layout(a.geometry, a.aliased, a.circuit)
            drow_layout = {"dots": (0, 3 * dots["ngroups"], Ld["cm1_2ns"][2])}
        if a.colmajor:                                    # (the one-row table programs address their own table: row-major as they are)
            Lc, ext = LW or G.layout(a.geometry, a.aliased, a.circuit), G.CIRCUITS[a.circuit]["ext"]
            geo = {st: ((G.to_colmajor(v[0], drow_layout if st == "step52ns_drow" else Lc, ext), v[1], v[2]) if not st.endswith("_tab") else v) for st, v in geo.items()}
    for size in sizes:
        for step in (x for x in a.steps.split(",") if x):
            lr = (24 if step.endswith("2ns") else 23) if size in ("full", "geo") else size
            src = os.path.join(R.REF_DIR, f"{a.circuit if geo else 'zkevm'}.{step.replace('_opt', '').replace('_tab', '').replace('_drow', '')}.xlpexpr")
            name = f"{a.circuit}.{step}.geo{a.geometry}{'a' if a.aliased else ''}{f'w{a.world}' if a.world else ''}{'c' if a.colmajor else ''}" if geo else f"zkevm.{step}.n{lr}"
            so = os.path.join(a.out_dir or R.REF_DIR, name + ".so")
            if not os.path.exists(src):
                print("missing", src); continue
            code, ntmp1, ntmp3 = R.read_program(src)
            for tn in a.tune:                                 # this step's own generator options (read when the text is written)
                nm, val = tn.split("=")
                if nm == step.replace("_opt", ""):
                    h, c = (int(x) for x in val.split(","))
                    os.environ["XLP_EXPR_HOIST_LOADS"] = str(h); os.environ["XLP_EXPR_CUT"] = str(c)
                    break
            else:
                for k, v in (("XLP_EXPR_HOIST_LOADS", a.hoist_loads), ("XLP_EXPR_CUT", a.cut)):
                    if v:
                        os.environ[k] = str(v)
                    else:
                        os.environ.pop(k, None)
            if geo:
                code, ntmp1, ntmp3 = geo[step]
            else:
                code, _ = R.shrink(code, 1 << lr)
            cm = bool(geo) and a.colmajor and not step.endswith("_tab")
            p = capi.ExprProgram(code, ntmp1, ntmp3, colmajor=cm, canonical=cm)      # (canonical: as host/steps.hpp creates the column-major programs)
            # regenerate the launcher text (cheap) and rebuild only if it changed: its signature covers the whole
            # compiled program, so a change of the library's compile step invalidates the plugin too
            build = os.path.join(a.out_dir or R.REF_DIR, "build")
            os.makedirs(build, exist_ok=True)
            probe = os.path.join(build, name + ".probe.hip")
            p.codegen(probe)
            stamp = os.path.join(build, name + ".stamp")
            # ... and the generated text is compiled against the library's field arithmetic: a change there (gl_field.hpp,
            # xlp_expr_jit.h) makes the plugin stale as well
            # ... and so does a change of the generator itself (segment size, lazy accumulation): the stamp covers the generated text
            import hashlib
            sig = [l for l in open(probe) if "xlpjit_signature" in l][0] + "headers " + headers_hash() + "\ntext " + hashlib.sha256(open(probe, "rb").read()).hexdigest()[:16] + "\n"
            if os.path.exists(so) and os.path.exists(stamp) and open(stamp).read() == sig and not a.force:
                print("up to date", so); continue
            t = time.time()
            p.compile(build, name, parts=a.parts)
            os.replace(os.path.join(build, name + ".so"), so)
            open(stamp, "w").write(sig)
            print(f"{so}: {time.time() - t:.1f} s", flush=True)
            # what the generator says the kernels move per row, and what the compiled code does about latency and registers (CPU: llvm-objdump)
            import json
            from tools import isa_loads_in_flight as ISA
            rep = {"generator": next((l.strip()[3:] for l in open(probe) if l.startswith("// xlp_expr_codegen:")), ""), "isa": ISA.summary(so)}
            json.dump(rep, open(so[:-3] + ".report.json", "w"), indent=1)
            print(f"{name}: {rep['generator']}\n{name}: {rep['isa']}", flush=True)


if __name__ == "__main__":
    main()
