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


BN128_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class HostTranscriptBN128:
    """TranscriptBN128 (transcriptBN128.cpp:5-141) of StarkRecursiveF: one Fr per put value, 16 pending + 1 state element per
    Poseidon-BN128 permutation (t = 17), challenges read as three 64-bit limbs per output element, query indices 253 bits per
    element.  perm17(list of 17 integers mod r) -> list of 17 integers: the permutation comes from the backend (the C++ mirror is
    `TranscriptBN128` in host/xlp_host_bn128.hpp)."""

    def __init__(self, perm17):
        self.perm = perm17
        self.state = [0]; self.pending = []; self.out = [0]; self.out3 = []

    def put(self, vals):
        for v in vals:
            self.pending.append(int(v) % BN128_R)
            self.out = []
            if len(self.pending) == 16:
                self._update()

    def _update(self):
        self.pending += [0] * (16 - len(self.pending))
        self.out = [int(v) for v in self.perm(self.out + self.state + self.pending)]
        self.state = [self.out[0]]; self.out3 = []; self.pending = []

    def get_fields253(self):
        while not self.out:
            self._update()
        return self.out.pop(0)

    def get1(self):
        while not self.out3:
            if not self.out:
                self._update()
            v = self.out.pop(0)
            self.out3 += [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(3)]
        return self.out3.pop(0)

    def get_field(self):
        return [self.get1() for _ in range(3)]

    def get_permutations(self, n, nbits):
        nf = (n * nbits - 1) // 253 + 1
        bits = []
        for _ in range(nf):
            f = self.get_fields253()
            bits += [(f >> b) & 1 for b in range(253)]
        return [sum(bits[i * nbits + j] << j for j in range(nbits)) for i in range(n)]
