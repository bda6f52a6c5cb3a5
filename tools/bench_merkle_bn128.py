"""MerkleTreeBN128::merkelize on one MI355X at the recursivef geometry (2^23 leaves x 12 Goldilocks columns = 4 Fr per leaf -> one t = 5
Poseidon-BN128 permutation per leaf, 559 241 t = 17 permutations for the 16-ary levels) and a wider section.  Prints one JSON line."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlayer_prover_amd import capi


def run(lg, w, reps=3):
    h = 1 << lg
    src = torch.empty(h * w, dtype=torch.int64, device="cuda")
    capi.dev_fill(src, 0x5EED0F04, 0)
    nn = capi.merkle_bn128_num_nodes(h)
    nodes = torch.empty(nn * 4, dtype=torch.int64, device="cuda")
    capi.dev_merkle_bn128(nodes, src, w, h); capi.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        capi.dev_merkle_bn128(nodes, src, w, h)
    capi.sync()
    dt = (time.perf_counter() - t0) / reps
    nfr = (w + 2) // 3
    leaf_perms = h * ((nfr + 15) // 16) if w > 4 else 0
    node_perms, n = 0, h
    while n > 1:
        n = (n + 15) // 16; node_perms += n
    return {"leaves": f"2^{lg} x {w}", "seconds": round(dt, 4), "leaf_permutations": leaf_perms, "leaf_t": min(nfr, 16) + 1, "node_permutations_t17": node_perms,
            "root_limbs": [int(v) & 0xFFFFFFFFFFFFFFFF for v in nodes[-4:].tolist()]}


def main():
    capi.init(0); capi.use_torch_stream()
    print(json.dumps({"workload": "MerkleTreeBN128::merkelize (StarkRecursiveF trees)", "runs": [run(23, 12), run(20, 12), run(20, 48), run(16, 48)]}))


if __name__ == "__main__":
    main()
