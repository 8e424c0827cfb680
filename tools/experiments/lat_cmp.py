import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnina_amd import capi
capi.init(0)
G = np.load("tests/golden/cnn_goldens.npz")
out = {}
for name in ["default2017", "crossdock_default2018", "dense", "dense_1_3"]:
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    s = capi.Scorer([name]); s.set_receptor(rec_xyz, rec_smt)
    a = s.score_batch(poses[:1], lig_smt)
    g = s.score_grad(poses[:1], lig_smt)
    out[name] = (a["pose"], a["affinity"], g["lig_grad"])
np.save(sys.argv[1], out, allow_pickle=True)
