"""The ROW-SHARDED C++ Starks (host/starks.hpp genProofSharded; SURVEY.md 8e, BASELINE north_star) against the one-GPU C++ Starks: the same
synthetic witness, constants and plookup selections must give the SAME proof, bit for bit (zkin text), whatever the number of ranks.
One process per rank (host/test_host_shim `bundle ... --world G --rank r`), exchanges through the library's transport (csrc/comm.hip):
  * world 1, backend "self"  -- the window layout and every exchange code path with device copies;
  * world 1, backend "rccl"  -- the same through RCCL (librccl loaded, communicator of one rank, all-gathers and ring steps on it);
  * world 2 / 4, backend "shm" -- real ranks, host-staged because they share the test box's one GPU (RCCL wants one GPU per rank).
More than one GPU has never been available to this build; what is checked here is that the multi-rank orchestration is correct."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "xlayer_prover_amd", "host", "test_host_shim")
WSEED, CSEED = 0x5EED0401, 0x5EED0402


def _programs_built():
    from xlayer_prover_amd import programs as R
    return os.path.exists(os.path.join(R.REF_DIR, "zkevm.step52ns.xlpexpr"))


def test_window_bundle_is_consistent(tmp_path):
    """no GPU: the per-rank bundle of a sharded proof -- window offsets / rows in the starkinfo, the same views as the one-GPU starkinfo,
    next-row stores listed for the programs that have them"""
    from tools import make_stark_bundle as B
    from tools import zkevm_geometry as G
    if not _programs_built():
        pytest.skip("converted reference programs not built")
    ln, world = 12, 4
    out = B.make("zkevm", ln, False, out=str(tmp_path / "w"), world=world)
    j = json.load(open(os.path.join(out, "zkevm.starkinfo.json")))
    one = G.synth_starkinfo("zkevm", ln, False)
    L = G.layout_window(ln, world)
    assert j["xlp"]["world"] == world and j["xlp"]["haloN"] == 1 and j["xlp"]["haloExt"] == 2
    assert j["starkStruct"] == one["starkStruct"] and j["varPolMap"] == one["varPolMap"] and j["evMap"] == one["evMap"] and j["puCtx"] == one["puCtx"]
    n, ne = (1 << ln) // world, (1 << (ln + 1)) // world
    for name, off in j["mapOffsets"].items():
        assert off == L[name][0] and j["mapDeg"][name] == (ne + 2 if name.endswith("_2ns") else n + 1)
    assert j["mapTotalN"] == L["end"][0] < one["mapTotalN"]
    man = json.load(open(os.path.join(out, "zkevm.steps.json")))["steps"]
    assert man["step2prev"]["nextRowStores"] == [] and man["step42ns"]["nextRowStores"] == []
    st = man["step3prev"]["nextRowStores"]
    assert st and all(s[0] in ("cm3_n", "tmpExp_n") and s[3] == 1 for s in st)
    W = {"cm3_n": 408, "tmpExp_n": 389}
    assert all(s[1] + s[2] <= W[s[0]] for s in st)


def _run_ranks(bundle, world, backend, tmp_path, timeout=300, env=None):
    # (timeouts stay below the 7 minutes of silence after which a GPU box takes a run for hung: a stuck rank then fails ONE test with its log)
    rdv = str(tmp_path / "rccl.id") if backend == "rccl" else f"xlp_test_{os.getpid()}_{world}_{abs(hash(str(tmp_path))) % 10**8}"    # a name of its own per launch
    procs = []
    for r in range(world):
        cmd = [EXE, "bundle", bundle, "zkevm", str(WSEED), str(CSEED), "--world", str(world), "--rank", str(r)]
        if backend != "self":
            cmd += ["--backend", backend, "--rendezvous", rdv]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(env or os.environ, XLP_COMM_BARRIER_TIMEOUT_S="200")))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: {se[-3000:]}"
    return json.loads(outs[0][0].strip().splitlines()[-1]), [se for _, se in outs]      # (RCCL prints a banner on stdout first)


_ONE_GPU = {}


def _one_gpu_proof(tmp_path, ln):
    """the one-GPU proof of the same witness and constants (one run per size serves every parametrisation)"""
    from tools import make_stark_bundle as B
    if ln not in _ONE_GPU:
        one = B.make("zkevm", ln, False, out=str(tmp_path / "one"))
        r = subprocess.run([EXE, "bundle", one, "zkevm", str(WSEED), str(CSEED)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        _ONE_GPU[ln] = json.loads(r.stdout)
    return json.loads(json.dumps(_ONE_GPU[ln]))


@pytest.mark.gpu
@pytest.mark.parametrize("world,backend,scratch_mb,msg_words,colmajor", [
    (2, "shm", 1, 777, False),                          # row-major windows: one case (the layout is kept for the equivalence test only)
    (1, "self", 0, 0, True), (2, "shm", 0, 0, True), (2, "shm", 1, 0, True), (4, "shm", 1, 777, True), (1, "rccl", 1, 1000, True)])
def test_cpp_sharded_starks_proof_equals_one_gpu_proof(tmp_path, world, backend, scratch_mb, msg_words, colmajor):
    """colmajor: the windows as COLUMNS (a column of a section = the rank's rows + halo; the extended ones as 2^ext coset blocks): what
    travels in the exchanges is regrouped runs of a rank's rows, the halo of a section is one row per column and coset.
    scratch_mb: the staging budget of the sharded commit (XLP_SHARD_SCRATCH_MB); 1 MiB forces the exchanges / LDE of a section into many
    column bands, i.e. the two-stream pipeline of commitSharded with both staging sets in use (at the real size a 2-rank zkEVM section needs
    14 bands to stay below 24 GB).  msg_words: XLP_COMM_MAX_MESSAGE_WORDS -- messages of an exchange / all-gather travel in ROUNDS of that many
    words (256 MiB in production, never reached at test sizes); with rccl in a world of one the rank's own block is sent through
    ncclSend / ncclRecv too (XLP_COMM_SELF_VIA_RCCL), so the grouped multi-round loop runs on RCCL itself."""
    from tools import make_stark_bundle as B
    if not _programs_built():
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    ln = 12
    expect = _one_gpu_proof(tmp_path, ln)
    win = B.make("zkevm", ln, False, out=str(tmp_path / "win"), world=world, colmajor=colmajor)
    env = dict(os.environ)
    if scratch_mb:
        env["XLP_SHARD_SCRATCH_MB"] = str(scratch_mb)
    if msg_words:
        env["XLP_COMM_MAX_MESSAGE_WORDS"] = str(msg_words)
        if backend == "rccl":
            env["XLP_COMM_SELF_VIA_RCCL"] = "1"
    got, logs = _run_ranks(win, world, backend, tmp_path, env=env)
    assert "exchanges" in logs[0]
    for k in ("root1", "root2", "root3", "root4"):
        assert got[k] == expect[k], k
    assert got["evals"] == expect["evals"]
    assert got == expect                                              # every FRI tree, every opened row and sibling path, the final polynomial


@pytest.mark.gpu
@pytest.mark.parametrize("colmajor", [True])
def test_genbatchproof_with_the_zkevm_stark_sharded_equals_one_gpu(tmp_path, colmajor):
    """Prover::genBatchProof (host/prover.hpp, binary host/zkprover_batch) with --world 2: both ranks prove their rows of the zkEVM STARK
    (window bundle, 2^13 rows), the publics come from rank 0's first and rank 1's last witness row, rank 0 goes on with c12a and
    recursive1 at their real geometry -- the batch proof it prints is the one-GPU run's, byte for byte (FNV of the zkin text)."""
    from tools import make_stark_bundle as B
    if not _programs_built():
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    exe = os.path.join(ROOT, "xlayer_prover_amd", "host", "zkprover_batch")
    small = {"c12a": B.make("c12a", 20, False, out=str(tmp_path / "c12a")), "recursive1": B.make("recursive1", 17, False, out=str(tmp_path / "recursive1"))}

    def config(name, zk):
        cfg = {"runFileGenBatchProof": True, "saveOutputToFile": False, "saveProofToFile": False, "outputPath": str(tmp_path), "xlpSyntheticLookups": True}
        for n, b in dict(small, zkevm=zk).items():
            cfg[n + "StarkInfo"] = os.path.join(b, f"{n}.starkinfo.json")
            cfg[n + "ConstPols"] = f"synthetic:{CSEED + len(n)}"
            cfg[n + "ConstantsTree"] = ""
            cfg[n + "Verkey"] = ""
        f = tmp_path / name
        f.write_text(json.dumps(cfg))
        return str(f)
    one = config("one.json", B.make("zkevm", 13, False, out=str(tmp_path / "zk1")))
    r = subprocess.run([exe, "-c", one, "--proofs", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    expect = json.loads(r.stdout.strip().splitlines()[-1])
    two = config("two.json", B.make("zkevm", 13, False, out=str(tmp_path / "zk2"), world=2, colmajor=colmajor))
    name = f"xlp_test_bp_{os.getpid()}"
    procs = [subprocess.Popen([exe, "-c", two, "--proofs", "1", "--world", "2", "--rank", str(k), "--backend", "shm", "--rendezvous", name],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for k, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {k}: {se[-3000:]}"
    got = json.loads(outs[0][0].strip().splitlines()[-1])
    assert outs[1][0].strip() == ""                                   # only rank 0 reports
    assert got["sharded"]["world"] == 2 and got["sharded"]["exchanges"] > 0
    assert got["recursive1_root1"] == expect["recursive1_root1"]
    assert got["batch_proof_fnv"] == expect["batch_proof_fnv"] and got["batch_proof_bytes"] == expect["batch_proof_bytes"]


@pytest.mark.gpu
@pytest.mark.parametrize("ln,colmajor", [(19, True)])
def test_genbatchproof_two_ranks_at_2p18_with_the_default_staging_budget(tmp_path, ln, colmajor):
    """the multi-rank product path at a size where the exchanges are hundreds of MB (2^19 rows: 1.6 GB of witness per rank, sections
    of 751 / 168 / 408 columns), through host/zkprover_batch with the DEFAULT staging budget and message size, ranks sharing the box's GPU (shm).
    Rank 0's line must name 2 ranks seen by the transport, the same zkEVM proof on both ranks, per-stage exchange volumes -- and the batch proof
    of the one-GPU run."""
    from tools import make_stark_bundle as B
    if not _programs_built():
        pytest.skip("converted reference programs not built (needs the reference tree at build time)")
    exe = os.path.join(ROOT, "xlayer_prover_amd", "host", "zkprover_batch")
    small = {"c12a": B.make("c12a", 20, False, out=str(tmp_path / "c12a")), "recursive1": B.make("recursive1", 17, False, out=str(tmp_path / "recursive1"))}

    def config(name, zk):
        cfg = {"runFileGenBatchProof": True, "saveOutputToFile": False, "saveProofToFile": False, "outputPath": str(tmp_path), "xlpSyntheticLookups": True,
               "zkevmCmPolsAfterExecutor": str(tmp_path / (name + ".commit"))}       # prover.cpp:452-457: every rank writes ITS rows of the same file
        for n, b in dict(small, zkevm=zk).items():
            cfg[n + "StarkInfo"] = os.path.join(b, f"{n}.starkinfo.json")
            cfg[n + "ConstPols"] = f"synthetic:{CSEED + len(n)}"
            cfg[n + "ConstantsTree"] = ""
            cfg[n + "Verkey"] = ""
        f = tmp_path / name
        f.write_text(json.dumps(cfg))
        return str(f)
    # (2^19 rows on COLUMN-MAJOR windows: 1.6 GB per rank through the exchanges of step 1, regrouped runs, the band-wise leaf sponge)
    one = config("one.json", B.make("zkevm", ln, False, out=str(tmp_path / "zk1")))
    r = subprocess.run([exe, "-c", one, "--proofs", "1"], capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr[-3000:]
    expect = json.loads(r.stdout.strip().splitlines()[-1])
    two = config("two.json", B.make("zkevm", ln, False, out=str(tmp_path / "zk2"), world=2, colmajor=colmajor))
    name = f"xlp_test_bp{ln}_{os.getpid()}"
    procs = [subprocess.Popen([exe, "-c", two, "--proofs", "1", "--timers", "--world", "2", "--rank", str(k), "--backend", "shm", "--rendezvous", name],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=360))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for k, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {k}: {se[-3000:]}"
    got = json.loads(outs[0][0].strip().splitlines()[-1])
    sh = got["sharded"]
    assert sh["world"] == 2 and sh["ranks_seen"] == 2 and sh["same_zkevm_proof_on_all_ranks"] is True
    st = sh["exchange_by_stage_rank0"]
    assert set(st) >= {"STARK_STEP_1_LDE_AND_MERKLETREE", "STARK_STEP_2_LDE_AND_MERKLETREE", "STARK_STEP_3_LDE_AND_MERKLETREE"}
    n = 1 << ln
    # step 1: my rows of the peer's columns out (n/2 x ~375 words), my columns of the peer's extended rows out (n x ~375 words): ~0.59 GB
    assert 0.9 * 8 * (n // 2 + n) * 375 < st["STARK_STEP_1_LDE_AND_MERKLETREE"]["GB_sent"] * 1e9 < 1.1 * 8 * (n // 2 + n) * 376
    assert got["batch_proof_fnv"] == expect["batch_proof_fnv"] and got["batch_proof_bytes"] == expect["batch_proof_bytes"]
    # the executor's output dumped by TWO concurrent ranks (each its rows, no rank truncating the other's: ADVICE r4) == the one-process dump, byte for byte
    a, b = open(tmp_path / "one.json.commit", "rb").read(), open(tmp_path / "two.json.commit", "rb").read()
    assert len(a) == n * 751 * 8 and a == b


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["geo12w2c"])
def test_cpp_sharded_starks_on_compiled_window_programs(tmp_path, tag):
    """two ranks on the AHEAD-OF-TIME kernels of the window-relocated step programs (the prebuilt bundle chelpers/bundles/zkevm.geo12w2, made by
    `make -C xlayer_prover_amd/chelpers bundles` like the 2^23 bundles bench.py --gpus N uses): every rank reports its four step programs as compiled,
    and the proof equals the one-GPU proof of the INTERPRETED programs"""
    from tools import make_stark_bundle as B
    from xlayer_prover_amd import programs as R
    win = os.path.join(R.REF_DIR, "bundles", "zkevm." + tag)        # (...c: the windows as columns)
    if not os.path.exists(os.path.join(win, "zkevm.steps.json")):
        pytest.skip("prebuilt window bundle not present (make -C xlayer_prover_amd/chelpers bundles)")
    man = json.load(open(os.path.join(win, "zkevm.steps.json")))["steps"]
    assert all(man[st]["plugin"] for st in ("step2prev", "step3prev", "step3", "step42ns")), "window plugins not built"
    one = B.make("zkevm", 12, False, out=str(tmp_path / "one"), use_opt=False, use_dots=False)       # the programs as converted, interpreted
    r = subprocess.run([EXE, "bundle", one, "zkevm", str(WSEED), str(CSEED)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "compiled" not in r.stderr
    expect = json.loads(r.stdout)
    got, logs = _run_ranks(win, 2, "shm", tmp_path)
    for log in logs:
        assert log.count("compiled") >= 4, log[-1500:]
    assert got == expect
