import os, sys, time, subprocess, json
# careful timing of whole extendPol per library: 3 rounds, each library in turn, 20 steps each (separate processes)
libs = sys.argv[1:]
code = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from xlayer_prover_amd import capi
capi.init(0); capi.use_torch_stream()
lg, ncols = 23, 8
n, n_ext = 1 << lg, 1 << (lg + 1)
src = torch.empty(n * ncols, dtype=torch.int64, device="cuda"); capi.dev_fill(src, 1, 0)
out = torch.empty(n_ext * ncols, dtype=torch.int64, device="cuda")
for _ in range(5): capi.dev_extend(out, src, n_ext, n, ncols)
capi.sync(); t0 = time.perf_counter()
for _ in range(40): capi.dev_extend(out, src, n_ext, n, ncols)
capi.sync(); wall = (time.perf_counter() - t0) / 40 * 1e3
capi.profile_enable(True)
for _ in range(5): capi.dev_extend(out, src, n_ext, n, ncols)
ms, cnt = capi.profile_read()
print(round(wall, 4), [round(x / 5, 4) for x in ms])
'''
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, XLP_LIB=os.path.abspath(l))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(os.path.basename(l), r.stdout.strip(), flush=True)
