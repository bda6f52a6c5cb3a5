"""The fflonk prover of the final proof (src/rapidsnark/fflonk_prover.c.hpp): host/fflonk_prover.hpp + csrc/fflonk.hip against
oracle/fflonk_oracle.py.

The reference snapshot holds no fflonk proving key: PARITY UNPINNED against the reference.  What is checked instead:
  * Keccak-256: the sponge against hashlib's SHA3-256 (NIST suffix), the original suffix against three published digests;
  * the oracle's prover on a synthetic key (random PLONK circuit, known tau) passes the fflonk verification equation (the pairing taken through
    the trapdoor), and a tampered proof / public input / witness does not -- CPU;
  * the GPU prover (one process: host/test_host_shim fflonk <zkey> <wtns> <blinding>) reads the same key and witness files in the reference's
    byte format and returns THE SAME proof as the oracle for the same blinding factors, bit for bit -- commitments through the multi-exponentiation
    over the key's powers of tau against the oracle's [p(tau)]_1 -- and with random blinding a proof that verifies;
  * the vector primitives (division by X^m - beta as a scan, running products, batch inverse, evaluation, divZh, interleave, linear combinations)
    against the oracle's sequential forms at sizes that take one, two and three scan levels."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "xlayer_prover_amd", "host", "test_host_shim")


def test_keccak256_against_hashlib_and_known_answers():
    """(1) the same sponge with the NIST suffix 0x06 IS SHA3-256: compared with hashlib on messages of 0 .. several blocks (rate 136), around
    the block boundary -- an independent implementation pins the permutation, the absorption and the padding; (2) what is left, the 0x01
    suffix of the original Keccak (keccak_wrapper.cpp:9), by the widely published Keccak-256 digests of "", "abc" and the fox sentence."""
    import hashlib
    from oracle import fflonk_oracle as F
    for n in (0, 1, 31, 135, 136, 137, 200, 271, 272, 1000):
        msg = bytes((7 * i + n) & 255 for i in range(n))
        assert F.keccak256(msg, suffix=0x06) == hashlib.sha3_256(msg).digest(), n
        assert F.keccak256(msg) != hashlib.sha3_256(msg).digest()
    assert F.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert F.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    assert F.keccak256(b"The quick brown fox jumps over the lazy dog").hex() == "4d741b6f1eb29cb2a9b9911c82f56fa8d73b04959d3d9d222895df6c0b28aa15"


def test_the_references_own_fflonk_proof_has_the_shape_this_prover_emits():
    """testvectors/finalProof/proof.json (the aggregator service test's mock output, test/service/aggregator/aggregator_service.cpp:56) is a real fflonk
    proof as 24 32-byte words; without its verification key nothing in it can be re-derived, but its SHAPE pins what a proof consists of: four G1
    commitments (on the curve: tests/test_bn128.py) and sixteen scalars below r -- the sixteen evaluations ql qr qm qo qc s1 s2 s3 a b c z zw t1w t2w inv
    that FflonkProver::prove adds to its proof (fflonk_prover.c.hpp:1217-1238,537), no more, no fewer"""
    from oracle import fflonk_oracle as F
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "fflonk_proof_g1_points.json")))
    assert len(g["points"]) == 4 and len(g["evaluations"]) == 16
    assert all(0 <= int(v) < F.P for v in g["evaluations"]) and len(set(g["evaluations"])) == 16
    z, wit, tau = F.setup(4, n_public=1, n_additions=0, seed=3)
    pj, _ = F.proof_to_json(*F.prove(z, wit, tau, list(range(1, 10)))[:2])
    assert sorted(pj["polynomials"]) == ["C1", "C2", "W1", "W2"]
    assert sorted(pj["evaluations"]) == sorted(["ql", "qr", "qm", "qo", "qc", "s1", "s2", "s3", "a", "b", "c", "z", "zw", "t1w", "t2w", "inv"])


def test_host_keccak_and_challenge_equal_the_oracles():
    """host/fflonk_prover.hpp without a GPU: keccak256() and the challenge derivation of Keccak256Transcript (digest as a big-endian integer
    mod r through the Montgomery product, back to bytes, a field inversion) against the oracle, on one- and many-block messages"""
    from oracle import fflonk_oracle as F
    if not os.path.exists(EXE):
        pytest.skip("host/test_host_shim not built")
    for n in (0, 1, 64, 135, 136, 137, 300):
        msg = bytes((11 * i + 3 * n) & 255 for i in range(n))
        r = subprocess.run([EXE, "keccak", msg.hex()], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr[-500:]
        j = json.loads(r.stdout)
        d = F.keccak256(msg)
        ch = int.from_bytes(d, "big") % F.P
        assert j["keccak256"] == d.hex() and j["challenge"] == str(ch) and j["challenge_be"] == ch.to_bytes(32, "big").hex(), n
        assert j["inverse_times_value"] == ("1" if ch else "0")


@pytest.mark.parametrize("power,npub,nadd,seed", [(4, 1, 0, 3), (5, 2, 3, 7), (6, 1, 5, 11)])
def test_oracle_proof_passes_the_verification_equation_and_tampering_does_not(power, npub, nadd, seed):
    from oracle import fflonk_oracle as F
    z, wit, tau = F.setup(power, n_public=npub, n_additions=nadd, seed=seed)
    proof, pub, _ = F.prove(z, wit, tau, [seed * 1000 + i for i in range(9)])
    pj, sj = F.proof_to_json(proof, pub)
    assert F.verify(z, tau, pj, sj)
    for key in ("a", "zw", "t2w", "s3", "inv"):
        bad = json.loads(json.dumps(pj))
        bad["evaluations"][key] = str((int(bad["evaluations"][key]) + 1) % F.P)
        assert not F.verify(z, tau, bad, sj), key
    bad = json.loads(json.dumps(pj))
    bad["polynomials"]["W2"] = bad["polynomials"]["W1"]
    assert not F.verify(z, tau, bad, sj)
    assert not F.verify(z, tau, pj, [str((int(sj[0]) + 1) % F.P)] + sj[1:])
    other = F.proof_to_json(*F.prove(z, wit, tau, F.random_blinding())[:2])
    assert other[0] != pj and F.verify(z, tau, *other)             # blinding changes the proof, not its validity
    wrong = list(wit)
    wrong[npub + 2] = (wrong[npub + 2] + 1) % F.P
    with pytest.raises((ValueError, AssertionError)):
        F.prove(z, wrong, tau, list(range(9)))


def test_zkey_and_wtns_files_have_the_sections_the_reference_reads(tmp_path):
    from oracle import fflonk_oracle as F
    z, wit, tau = F.setup(4, n_public=2, n_additions=2, seed=5)
    F.write_zkey(str(tmp_path / "k.zkey"), z)
    F.write_wtns(str(tmp_path / "w.wtns"), wit)
    s = F.read_binfile(str(tmp_path / "k.zkey"), "zkey")
    n = z["n"]
    assert sorted(s) == list(range(1, 18)) and s[1] == (10).to_bytes(4, "little")
    assert len(s[F.SEC["ql"]]) == 5 * n * 32 and len(s[F.SEC["c0"]]) == 8 * n * 32 and len(s[F.SEC["ptau"]]) == 9 * n * 64
    assert len(s[F.SEC["lagrange"]]) == 2 * 5 * n * 32 and len(s[F.SEC["additions"]]) == 2 * 72 and len(s[F.SEC["amap"]]) == 4 * z["n_constraints"]
    assert F.fr_from_mont_bytes(s[F.SEC["s2"]][:n * 32]) == z["pols"]["s2"]
    assert F.g1_from_bytes(s[F.SEC["ptau"]], 1) == F.g1_mul(F.G1, tau)
    w = F.read_binfile(str(tmp_path / "w.wtns"), "wtns")
    assert int.from_bytes(w[1][36:40], "little") == len(wit) and int.from_bytes(w[2][32:64], "little") == wit[1]


def _run(tmp_path, z, wit, blinding, env=None):
    from oracle import fflonk_oracle as F
    F.write_zkey(str(tmp_path / "k.zkey"), z)
    F.write_wtns(str(tmp_path / "w.wtns"), wit)
    args = [EXE, "fflonk", str(tmp_path / "k.zkey"), str(tmp_path / "w.wtns")]
    if blinding is not None:
        with open(tmp_path / "b.bin", "wb") as f:
            for v in blinding:
                f.write(int(v).to_bytes(32, "little"))
        args.append(str(tmp_path / "b.bin"))
    r = subprocess.run(args, capture_output=True, text=True, timeout=360, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("power,npub,nadd,seed", [(4, 1, 0, 3), (7, 2, 4, 21), (9, 1, 7, 22)])
def test_gpu_prover_returns_the_oracles_proof(tmp_path, power, npub, nadd, seed):
    from oracle import fflonk_oracle as F
    z, wit, tau = F.setup(power, n_public=npub, n_additions=nadd, seed=seed)
    blinding = [F.Rng(seed + 77).fr() for _ in range(9)]
    expect = F.proof_to_json(*F.prove(z, wit, tau, blinding)[:2])
    got = _run(tmp_path, z, wit, blinding)
    assert got["publics"] == expect[1]
    for k in ("C1", "C2", "W1", "W2"):
        assert got["proof"]["polynomials"][k] == expect[0]["polynomials"][k], k
    assert got["proof"]["evaluations"] == expect[0]["evaluations"]
    assert got["proof"] == expect[0]
    assert F.verify(z, tau, got["proof"], got["publics"])
    rnd = _run(tmp_path, z, wit, None)                               # blinding drawn by the prover: another proof, as valid
    assert rnd["proof"] != got["proof"] and F.verify(z, tau, rnd["proof"], rnd["publics"])


@pytest.mark.gpu
def test_gpu_prover_commitments_in_several_multiexponentiation_calls(tmp_path):
    """a commitment over more points than one xlp_dev_msm_bn254 call takes (2^26 in production: 9 x 2^24 points = three calls) is the sum of
    partial sums, added by one more multi-exponentiation with unit scalars: forced here with a chunk of 500 points (XLP_FFLONK_MSM_CHUNK)"""
    from oracle import fflonk_oracle as F
    z, wit, tau = F.setup(8, n_public=1, n_additions=3, seed=33)
    blinding = [F.Rng(133).fr() for _ in range(9)]
    expect = F.proof_to_json(*F.prove(z, wit, tau, blinding)[:2])
    got = _run(tmp_path, z, wit, blinding, env={"XLP_FFLONK_MSM_CHUNK": "500"})
    assert got["proof"] == expect[0] and got["publics"] == expect[1]


@pytest.mark.gpu
def test_gpu_prover_refuses_a_witness_that_breaks_the_copy_constraints(tmp_path):
    from oracle import fflonk_oracle as F
    z, wit, tau = F.setup(5, n_public=1, n_additions=2, seed=9)
    wrong = list(wit)
    wrong[3] = (wrong[3] + 1) % F.P
    F.write_zkey(str(tmp_path / "k.zkey"), z)
    F.write_wtns(str(tmp_path / "w.wtns"), wrong)
    r = subprocess.run([EXE, "fflonk", str(tmp_path / "k.zkey"), str(tmp_path / "w.wtns")], capture_output=True, text=True, timeout=360)
    assert r.returncode != 0 and ("not well calculated" in r.stderr or "Copy constraints" in r.stderr), r.stderr[-500:]


# ---------------------------------------------------------------------------------------------------------------- the vector primitives
@pytest.fixture(scope="module")
def xlp():
    import torch
    from xlayer_prover_amd import capi
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.init(0)
    capi.use_torch_stream()
    return capi


def _limbs(vals):
    from oracle.bn128_oracle import to_mont_limbs
    return np.array([to_mont_limbs(v) for v in vals], dtype=np.uint64).reshape(-1, 4)


def _vals(arr):
    from oracle.bn128_oracle import from_mont_limbs
    return [from_mont_limbs(r) for r in np.asarray(arr, dtype=np.uint64).reshape(-1, 4)]


@pytest.mark.gpu
@pytest.mark.parametrize("length,m", [(40, 1), (129, 3), (1000, 4), (20000, 8), (16400, 1), (5, 8)])
def test_division_by_xm_minus_beta_as_a_scan(xlp, length, m):
    """Polynomial::divByZerofier on a polynomial that IS divisible: product of a random quotient with X^m - beta"""
    import torch
    from oracle import fflonk_oracle as F
    rng = F.Rng(length * 31 + m)
    beta = rng.fr()
    q = [rng.fr() for _ in range(max(length - m, 0))]
    p = [0] * length
    for i, v in enumerate(q):                                        # (X^m - beta) q
        p[i + m] = (p[i + m] + v) % F.P
        p[i] = (p[i] - beta * v) % F.P
    want = F.div_by_zerofier(p, m, beta) if length > m else [0] * length
    assert want[:len(q)] == q
    src = xlp.from_np(_limbs(p).reshape(-1))
    out = torch.full((4 * length,), 7, dtype=torch.int64, device="cuda")
    xlp.dev_fr_div_xm_minus(out, src, length, m, _limbs([beta])[0])
    xlp.sync()
    assert _vals(xlp.to_np(out)) == [v % F.P for v in want]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 100, 128, 129, 16385, 40000])
def test_prefix_product_batch_inverse_and_evaluation(xlp, n):
    import torch
    from oracle import fflonk_oracle as F
    rng = F.Rng(n + 5)
    a = [rng.fr() or 1 for _ in range(n)]
    src = xlp.from_np(_limbs(a).reshape(-1))
    out = torch.zeros(4 * n, dtype=torch.int64, device="cuda")
    xlp.dev_fr_prefix_product(out, src, n)
    xlp.sync()
    run, acc = [], 1
    for v in a:
        acc = acc * v % F.P
        run.append(acc)
    assert _vals(xlp.to_np(out)) == run
    xlp.dev_fr_batch_inverse(out, src, n)
    xlp.sync()
    assert _vals(xlp.to_np(out)) == [F.inv(v) for v in a]
    pts = [rng.fr() for _ in range(5)] + [0, 1]
    got = xlp.dev_fr_eval(src, n, _limbs(pts))
    assert _vals(got) == [F.poly_eval(a, x) for x in pts]
    xlp.dev_fr_from_mont(out, src, n)
    xlp.sync()
    raw = xlp.to_np(out).reshape(-1, 4)
    assert [sum(int(r[i]) << (64 * i) for i in range(4)) for r in raw] == a


@pytest.mark.gpu
def test_div_zh_interleave_and_linear_combinations(xlp):
    import torch
    from oracle import fflonk_oracle as F
    rng = F.Rng(99)
    n = 64
    q = [rng.fr() for _ in range(3 * n - 5)] + [0] * 5
    p = [0] * (4 * n)
    for i, v in enumerate(q):                                        # (X^n - 1) q
        p[i + n] = (p[i + n] + v) % F.P
        p[i] = (p[i] - v) % F.P
    d = xlp.from_np(_limbs(p).reshape(-1))
    xlp.dev_fr_div_zh(d, 4 * n, n)
    xlp.sync()
    assert _vals(xlp.to_np(d)) == F.div_zh(p, n) and _vals(xlp.to_np(d))[:3 * n] == q
    pols = [[rng.fr() for _ in range(k)] for k in (10, 7, 12)]
    devs = [xlp.from_np(_limbs(c).reshape(-1)) for c in pols]
    out = torch.full((4 * 40,), 3, dtype=torch.int64, device="cuda")
    xlp.dev_fr_interleave(out, 40, [(devs[0], 10), (devs[1], 7), (devs[2], 12)])
    xlp.sync()
    want = [0] * 40
    for j, c in enumerate(pols):
        for i, v in enumerate(c):
            if 3 * i + j < 40:
                want[3 * i + j] = v
    assert _vals(xlp.to_np(out)) == want
    ks = [rng.fr(), rng.fr(), F.P - 1]
    xlp.dev_fr_lincomb(out, 12, [(devs[0], 0, 10, _limbs([ks[0]])[0]), (devs[1], 2, 5, _limbs([ks[1]])[0]), (devs[2], 0, 12, _limbs([ks[2]])[0])])
    xlp.sync()
    want = [((pols[0][i] * ks[0] if i < 10 else 0) + (pols[1][2 + i] * ks[1] if i < 5 else 0) + pols[2][i] * ks[2]) % F.P for i in range(12)]
    assert _vals(xlp.to_np(out))[:12] == want


@pytest.mark.gpu
def test_scratch_slots_can_be_lent_and_released(xlp):
    """xlp_scratch_lend / xlp_scratch_release (include/xlp_gl.h): a scan whose run totals need scratch slot 4 works in a region lent by the
    caller (and writes there), falls through to the library's own scratch when the region is too small, and again after everything was released"""
    import torch
    from oracle import fflonk_oracle as F
    n = 5000
    rng = F.Rng(4040)
    a = [rng.fr() or 1 for _ in range(n)]
    run, acc = [], 1
    for v in a:
        acc = acc * v % F.P
        run.append(acc)
    src = xlp.from_np(_limbs(a).reshape(-1))
    out = torch.zeros(4 * n, dtype=torch.int64, device="cuda")
    region = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    xlp.scratch_lend(4, region)
    xlp.dev_fr_prefix_product(out, src, n)
    xlp.sync()
    assert _vals(xlp.to_np(out)) == run and int((region != 0).sum()) > 0          # the run totals went through the lent region
    tiny = torch.zeros(8, dtype=torch.int64, device="cuda")
    xlp.scratch_lend(4, tiny)                                                       # too small: the slot's own scratch serves
    out.zero_()
    xlp.dev_fr_prefix_product(out, src, n)
    xlp.sync()
    assert _vals(xlp.to_np(out)) == run and int((tiny != 0).sum()) == 0
    xlp.scratch_lend(4, None)
    xlp.scratch_release(1 << 4)
    out.zero_()
    xlp.dev_fr_prefix_product(out, src, n)
    xlp.sync()
    assert _vals(xlp.to_np(out)) == run
