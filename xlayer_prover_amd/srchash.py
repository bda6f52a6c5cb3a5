"""Hash of the sources of the kernels whose hardware counters bench.py quotes (the extendPol kernels and the Poseidon leaf sponge):
rocprof summaries under profiles/ are stamped with it, and bench.py marks a counter-derived figure stale when the kernels it was
measured on are no longer the ones in the tree.  Host-side files of csrc/ (the C ABI, the transport) do not enter it; the compiler-flag
lines of csrc/Makefile do (the per-object scheduler strategies change the measured kernels' instruction streams)."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
MEASURED = ("gl_field.hpp", "ntt.hip", "ntt_common.hpp", "ntt_lde.hip", "ntt_lde_mid.hip", "ntt_lde_mid_kernel.inc", "ntt_lde_mid_mfma.hip", "ntt_mfma_common.hpp", "ntt_lde_common.hpp", "ntt_lde_cm.hip", "ntt_lde_cm_mid.hip", "ntt_lde_cm_common.hpp",
            "poseidon.hip", "poseidon_constants.inc", "poseidon_fused.inc")


def csrc_hash():
    h = hashlib.sha256()
    for name in MEASURED:
        h.update(name.encode()); h.update(b"\0")
        h.update(open(os.path.join(CSRC, name), "rb").read())
    for line in open(os.path.join(CSRC, "Makefile")):
        if "FLAGS" in line and not line.lstrip().startswith("#"):
            h.update(line.strip().encode()); h.update(b"\0")
    return h.hexdigest()[:16]
