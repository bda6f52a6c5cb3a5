"""xlayer_prover_amd -- MI355X-native STARK hot path (Goldilocks NTT/LDE,
Poseidon Merkle commitment, FRI folding) behind the C ABI in include/xlp_gl.h.

The product is libxlp_gl.so (HIP, gfx950).  This Python package is only the
thin ctypes plumbing used by tests/ and bench.py; the reference-shaped host
interface for C++ callers lives in xlayer_prover_amd/host/.
"""
from . import capi  # noqa: F401
from .capi import XlpError, lib_path  # noqa: F401
