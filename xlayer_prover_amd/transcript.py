"""Fiat-Shamir transcript of the STARK prover on the host (Python drivers); the C++ mirror is `Transcript` in host/xlp_host.hpp."""


class HostTranscript:
    """Transcript (transcript.cpp:4-87); the permutation comes from the backend"""

    def __init__(self, perm):
        self.perm = perm
        self.state = [0] * 4; self.pending = [0] * 8; self.out = [0] * 12; self.pc = 0; self.oc = 0

    def _absorb(self):
        self.out = [int(v) for v in self.perm(self.pending + self.state)]
        self.oc = 12; self.pending = [0] * 8; self.pc = 0; self.state = self.out[:4]

    def put(self, vals):
        for v in vals:
            self.pending[self.pc] = int(v); self.pc += 1; self.oc = 0
            if self.pc == 8:
                self._absorb()

    def get1(self):
        if self.oc == 0:
            self._absorb()
        r = self.out[(12 - self.oc) % 12]; self.oc -= 1
        return r

    def get_field(self):
        return [self.get1() for _ in range(3)]

    def get_permutations(self, n, nbits):
        nf = (n * nbits - 1) // 63 + 1
        bits = []
        for _ in range(nf):
            f = self.get1()
            bits += [(f >> b) & 1 for b in range(63)]
        return [sum(bits[i * nbits + j] << j for j in range(nbits)) for i in range(n)]
