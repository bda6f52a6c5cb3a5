#!/usr/bin/env python3
"""Per code object (a compiled step-program plugin, or any .so with a gfx950 offload bundle): how many global / scratch loads are in flight at each
`s_waitcnt vmcnt` that waits for a load, and how far ahead of that wait the YOUNGEST of them was issued (instructions).  CPU only (llvm-objdump).
What found the exposed memory latency of the generated step programs in round 4 (1.4 loads per wait, the youngest 10 instructions ahead), and what
tools/build_expr_plugins.py prints per plugin since the generator pipelines the loads itself (csrc/expr.hip place_loads).
    python3 tools/isa_loads_in_flight.py xlayer_prover_amd/chelpers/zkevm.step42ns_opt.geo23ac.so"""
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def code_objects(path):
    """the AMDGPU ELF images embedded in a host shared object (e_machine 224)"""
    d = open(path, "rb").read()
    i = 0
    while True:
        i = d.find(b"\x7fELF", i)
        if i < 0:
            return
        if i > 0 and struct.unpack_from("<H", d, i + 18)[0] == 224:
            shoff = struct.unpack_from("<Q", d, i + 40)[0]
            shentsize, shnum = struct.unpack_from("<HH", d, i + 58)
            yield d[i:i + shoff + shentsize * shnum]
        i += 4


def analyze(path):
    """-> dict(kernels, loads, waits, loads_per_wait, loads_in_flight, median_distance, mean_distance) over all kernels of the file"""
    loads = waits = nk = 0
    per_wait, dist, inflight = [], [], []
    for img in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(img); f.flush()
            dis = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True).stdout
        pend, n = [], 0
        for line in dis.splitlines():
            if not line.startswith("\t"):
                if line.endswith(">:"):
                    nk += 1
                pend = []
                continue
            n += 1
            if "global_load" in line or "scratch_load" in line:
                pend.append(n); loads += 1
            elif "global_store" in line or "scratch_store" in line:
                pend.append(None)
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", line)
            if m:
                k = int(m.group(1))
                if len(pend) > k:
                    nfl = sum(1 for x in pend if x is not None)
                    done, pend = pend[:len(pend) - k], pend[len(pend) - k:]
                    ld = [x for x in done if x is not None]
                    if ld:
                        waits += 1
                        per_wait.append(len(ld)); dist.append(n - max(ld)); inflight.append(nfl)
    dist.sort()
    # loads_per_wait: loads a wait retires; loads_in_flight: loads issued and not yet waited for when such a wait begins (what hides the latency:
    # a pipelined kernel retires its loads one at a time -- vmcnt(N), N > 0 -- with the younger ones still on their way)
    return {"kernels": nk, "loads": loads, "waits": waits, "loads_per_wait": round(sum(per_wait) / max(1, len(per_wait)), 2),
            "loads_in_flight": round(sum(inflight) / max(1, len(inflight)), 2),
            "median_distance": dist[len(dist) // 2] if dist else 0, "mean_distance": round(sum(dist) / max(1, len(dist)), 1)}


def registers(path):
    """-> per kernel (name, vector registers incl. accumulation registers, scratch bytes) from the code objects' metadata notes"""
    out = []
    for img in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(img); f.flush()
            notes = subprocess.run([OBJDUMP.replace("objdump", "readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s+(?:- )?\.(agpr_count|vgpr_count|private_segment_fixed_size|name):\s+(\S+)", line)
            if not m:
                continue
            if m.group(1) == "agpr_count" and cur:
                out.append(cur); cur = {}
            cur[m.group(1)] = m.group(2)
        if cur:
            out.append(cur)
    # (.vgpr_count is the unified total on gfx90a and later: it includes the accumulation registers)
    return [(k.get("name", "?"), int(k.get("vgpr_count", 0)), int(k.get("private_segment_fixed_size", 0))) for k in out]


def summary(path):
    """analyze() + the register picture: what tools/build_expr_plugins.py prints and writes next to a plugin"""
    r = analyze(path)
    regs = sorted(v for _, v, _ in registers(path))
    if regs:
        r.update({"vgprs_median": regs[len(regs) // 2], "vgprs_p90": regs[len(regs) * 9 // 10], "vgprs_max": regs[-1],
                  "kernels_over_128_vgprs": sum(1 for v in regs if v > 128), "kernels_over_256_vgprs": sum(1 for v in regs if v > 256),
                  "scratch_bytes_max": max(s for _, _, s in registers(path))})
    return r


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(p, summary(p))
