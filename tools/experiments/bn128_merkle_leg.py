import torch, json, sys
sys.path.insert(0, ".")
import bench
from xlayer_prover_amd import capi
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
capi.init(0); capi.use_torch_stream()
print(json.dumps(bench.bn128_leg(capi, torch, dev)["merkle_bn128"]))
