import os, subprocess, sys, json
sys.path.insert(0, os.getcwd())
from tools import make_stark_bundle as B
EXE = "xlayer_prover_amd/host/test_host_shim"
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
win = B.make("zkevm", 12, False, out="/tmp/dbg_win", world=world)
env = dict(os.environ, AMD_SERIALIZE_KERNEL="3", AMD_SERIALIZE_COPY="3", XLP_COMM_TRACE="1")
procs = [subprocess.Popen([EXE, "bundle", win, "zkevm", "0x5EED0401", "0x5EED0402", "timers", "--world", str(world), "--rank", str(r), "--backend", "shm", "--rendezvous", f"xlp_dbg_{os.getpid()}"],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
for r, p in enumerate(procs):
    try:
        so, se = p.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        p.kill(); so, se = p.communicate()
    print(f"==== rank {r} rc={p.returncode} stdout {len(so)} bytes; stderr tail:")
    print(se[-2500:])
