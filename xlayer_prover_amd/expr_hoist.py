"""Software pipelining of the column loads of a generated step-program kernel (a pass over the text xlp_expr_codegen writes, applied by
ExprProgram.compile when asked: XLP_EXPR_HOIST_LOADS=<batch>, tools/build_expr_plugins.py --hoist-loads).

Why: hipcc issues each global load of a generated segment right before its first use -- 1.4 loads per wait, the youngest load a median of 10
instructions before the wait (tools/experiments/isa_loads_in_flight.py on zkevm.step42ns) -- so a wavefront exposes the whole memory latency
~60 times per segment and only the 2-5 wavefronts per SIMD hide it: the segments ran at 2 TB/s and 30 % vector issue, neither pipe full.
What: the RAW loads of a kernel (T[...], PR[...], KN[...], KE[...], C[...]) are issued in batches, each one batch AHEAD of its use and pinned
by __builtin_amdgcn_sched_barrier(0); the canonicalisation stays at the place of use.  Legal because every load of a segment precedes every
store of it (checked per kernel: one that does not comply is left as it was).  Same values, same proof (zkEVM step42ns at 2^23 rows:
0.378 -> 0.302 s, batch proof FNV unchanged: profiles/r4/steps_hoist_loads_ab_*.json)."""
import re

_LOAD = re.compile(r'^(\s*)const u64 (v\d+) = (gl::canon\()?((?:T|PR|KN|KE|C)\[.*\])\)?;$')
_STORE = re.compile(r'^\s*(?:T|PW|CW|A\.P\.[A-Za-z_0-9]+)\[.*\]\s*=[^=]')


def _kernel(text, batch):
    lines = text.split('\n')
    loads = []
    first_store = None
    for i, l in enumerate(lines):
        m = _LOAD.match(l)
        if m:
            if first_store is not None:
                return text                                   # a load after a store: not this pass's kind of kernel
            loads.append((i, m.group(2), bool(m.group(3)), m.group(4), m.group(1)))
        elif first_store is None and _STORE.match(l):
            first_store = i
    if len(loads) < 2 * batch:
        return text
    batches = [loads[k:k + batch] for k in range(0, len(loads), batch)]
    emit_at = {}                                              # line -> batches whose raw loads go in front of it
    for bi in range(len(batches)):
        emit_at.setdefault(loads[0][0] if bi <= 1 else batches[bi - 1][0][0], []).append(bi)
    by_line = {ld[0]: ld for ld in loads}
    out = []
    for i, l in enumerate(lines):
        for bi in emit_at.get(i, ()):
            for (_, name, _c, expr, ind) in batches[bi]:
                out.append(f"{ind}const u64 {name}_raw = {expr};")
            out.append("    __builtin_amdgcn_sched_barrier(0);")
        if i in by_line:
            _, name, canon, _e, ind = by_line[i]
            out.append(f"{ind}const u64 {name} = gl::canon({name}_raw);" if canon else f"{ind}const u64 {name} = {name}_raw;")
        else:
            out.append(l)
    return '\n'.join(out)


def transform(src, batch=8):
    parts = re.split(r'(?=extern "C" __global__)', src)
    return parts[0] + ''.join(_kernel(p, batch) for p in parts[1:])
