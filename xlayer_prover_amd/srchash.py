"""Hash of the kernel sources (xlayer_prover_amd/csrc): rocprof summaries under profiles/ are stamped with it, and bench.py
marks a counter-derived figure stale when the kernels it was measured on are no longer the ones in the tree."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def csrc_hash():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".hpp", ".h", ".inc")):
            h.update(name.encode()); h.update(b"\0")
            h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()[:16]
