#!/usr/bin/env python3
"""Synthetic genProof (BASELINE.json configs[3] stand-in): the commit stages and FRI of
Starks::genProof (src/starkpil/starks.cpp:9-404) at the fork-9 zkEVM geometry (N = 2^23, N_ext = 2^24,
committed widths 751 / 168 / 408 / 6) on ONE MI355X, GPU-resident, in the reference's
stage order and with its timer names.  The witness and the constraint-evaluation outputs are synthetic
(splitmix64): the executor and the chelpers VM are outside this round's scope, so
this measures the NTT / Merkle / FRI share of a batch proof, not a full proof.

One section at a time is resident (input 50 GB + extension 101 GB + scratch 13 GB for cm1).
  python tools/bench_genproof.py [--log-n 23] [--widths 751,168,408,6] [--fri-steps 24,19,14,10,6] [--queries 128]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xlayer_prover_amd import capi


class HostTranscript:
    """Transcript (transcript.cpp:4-87) on the host; the permutation runs on the GPU (xlp_perm)."""

    def __init__(self):
        self.state = [0] * 4; self.pending = [0] * 8; self.out = [0] * 12; self.pc = 0; self.oc = 0

    def _absorb(self):
        o = capi.perm(self.pending + self.state)
        self.out = [int(v) for v in o]; self.oc = 12; self.pending = [0] * 8; self.pc = 0; self.state = self.out[:4]

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
        fields = [self.get1() for _ in range(nf)]
        bits = []
        for f in fields:
            bits += [(f >> b) & 1 for b in range(63)]
        return [sum(bits[i * nbits + j] << j for j in range(nbits)) for i in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=23)
    ap.add_argument("--widths", default="751,168,408,6")
    ap.add_argument("--fri-steps", default="24,19,14,10,6")
    ap.add_argument("--queries", type=int, default=128)
    args = ap.parse_args()
    P = 0xFFFFFFFF00000001
    N, NE = 1 << args.log_n, 1 << (args.log_n + 1)
    widths = [int(x) for x in args.widths.split(",")]
    steps = [int(x) for x in args.fri_steps.split(",")]
    assert steps[0] == args.log_n + 1
    capi.init(0); capi.use_torch_stream()
    dev = "cuda"
    wmax = max(widths[:3])
    t_alloc = time.perf_counter()
    cm_n = torch.empty(N * wmax, dtype=torch.int64, device=dev)
    cm_2ns = torch.empty(NE * wmax, dtype=torch.int64, device=dev)
    nodes = torch.empty((2 * NE - 1) * 4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    timers = {}

    def timed(name, fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        timers[name] = timers.get(name, 0.0) + time.perf_counter() - t0

    tr = HostTranscript()
    tr.put([1, 2, 3, 4]); tr.put(list(range(48)))            # verkey, publics (synthetic)
    t_proof = time.perf_counter()
    roots = []
    # ---- steps 1-3: extendPol + merkelize of cm1 / cm2 / cm3 (starks.cpp:53-60,134-141,215-222)
    for si, w in enumerate(widths[:3]):
        src = cm_n[: N * w]; ext = cm_2ns[: NE * w]
        capi.dev_fill(src, 0x5EED0040 + si, 0)                # synthetic witness / H1H2 / Z columns
        timed(f"STARK_STEP_{si+1}_LDE", lambda: capi.dev_extend(ext, src, NE, N, w))
        timed(f"STARK_STEP_{si+1}_MERKLETREE", lambda: capi.dev_merkle(nodes, ext, w, NE))
        root = capi.to_np(nodes[-4:]); roots.append([int(v) for v in root])
        tr.put(roots[-1]); tr.get_field(); tr.get_field() if si < 2 else None
    # ---- step 4: q polynomial: INTT (qDim 3), coset re-split is elementwise (not timed), NTT (6 cols), tree
    w4 = widths[3]
    q = cm_n[: NE * 3]; capi.dev_fill(q, 0x5EED0044, 0)
    qq = cm_2ns[: NE * 3]
    timed("STARK_STEP_4_CALCULATE_EXPS_2NS_INTT", lambda: capi.dev_ntt(qq, q, NE, 3, inverse=True))
    q6 = cm_n[: NE * w4]; capi.dev_fill(q6, 0x5EED0045, 0)
    cm4 = cm_2ns[: NE * w4]
    timed("STARK_STEP_4_CALCULATE_EXPS_2NS_NTT", lambda: capi.dev_ntt(cm4, q6, NE, w4))
    timed("STARK_STEP_4_MERKLETREE", lambda: capi.dev_merkle(nodes, cm4, w4, NE))
    roots.append([int(v) for v in capi.to_np(nodes[-4:])]); tr.put(roots[-1]); tr.get_field()
    # ---- step 5: LEv / LpEv INTTs (2 x 2^23 x 3)
    lev = cm_n[: N * 3]; capi.dev_fill(lev, 0x5EED0046, 0)
    timed("STARK_STEP_5_LEv_LpEv", lambda: (capi.dev_ntt(lev, lev, N, 3, inverse=True), capi.dev_ntt(lev, lev, N, 3, inverse=True)))
    tr.put(list(range(100, 100 + 3 * 118))); tr.get_field(); tr.get_field()
    # ---- FRI (friProve.cpp:5-190) on a synthetic f_2ns
    def fri():
        pol = cm_n[: NE * 3]; capi.dev_fill(pol, 0x5EED004F, 0)
        pol2 = cm_2ns[: NE * 3]; aux = cm_2ns[NE * 3: NE * 6]
        bits = steps[0]; sinv = pow(7, P - 2, P)
        for si, nb in enumerate(steps):
            red = bits - nb
            x = tr.get_field()
            capi.dev_fri_fold(pol2, pol, bits, red, x, sinv)
            if si < len(steps) - 1:
                ng = 1 << steps[si + 1]; gs = (1 << nb) // ng
                capi.dev_transpose3(aux, pol2, 1 << nb, steps[si + 1])
                capi.dev_merkle(nodes, aux, gs * 3, ng)
                tr.put([int(v) for v in capi.to_np(nodes[(2 * ng - 2) * 4: (2 * ng - 1) * 4])])
            else:
                fin = capi.to_np(pol2[: 3 << nb]); tr.put([int(v) for v in fin])
            pol, pol2 = pol2, pol
            bits = nb
            for _ in range(red):
                sinv = sinv * sinv % P
        return tr.get_permutations(args.queries, steps[0])
    timed("STARK_STEP_FRI", fri)
    total = time.perf_counter() - t_proof
    perms = sum(NE * ((w + 7) // 8) + NE - 1 for w in widths)
    out = {
        "workload": f"synthetic genProof commit+FRI stages, N=2^{args.log_n}, widths {widths}, FRI steps {steps}, {args.queries} queries, 1 GPU",
        "timers_s": {k: round(v, 4) for k, v in timers.items()},
        "sum_timed_s": round(sum(timers.values()), 4), "wall_s_incl_fill_and_transcript": round(total, 4),
        "lde_input_elements": N * sum(widths[:3]), "poseidon_permutations": perms,
        "roots": roots,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
