import sys, json, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if os.environ.get("WITH_TORCH"):
    import torch
    torch.zeros(4, device="cuda:0")
import bench
from gnina_amd import capi, synth
capi.init(0)
r = bench.config_seam_b1(capi, synth)
print(os.environ.get("WITH_TORCH"), json.dumps({k: r[k] for k in ("default2017", "default_ensemble")}))
