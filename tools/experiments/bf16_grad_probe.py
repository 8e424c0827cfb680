import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnina_amd import capi
capi.init(0)
G = np.load("tests/golden/cnn_goldens.npz")
for name, kw in (("default2017", {}), ("dense", {}), ("dense_1_3", {}), ("dense_1_3", dict(resolution=0.25, dimension=23.75))):
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    s = capi.Scorer([capi.Model(name, **kw)]); s.set_receptor(rec_xyz, rec_smt)
    a = s.score_grad(poses, lig_smt)
    s.set_precision(True)
    b = s.score_grad(poses, lig_smt)
    for i in range(len(poses)):
        ga, gb = a["lig_grad"][i], b["lig_grad"][i]
        print(name, kw, i, "loss", a["loss"][i], b["loss"][i], "max|g|", np.abs(ga).max(), "rel", np.abs(ga-gb).max()/np.abs(ga).max(), "cos", (ga*gb).sum()/np.linalg.norm(ga)/np.linalg.norm(gb))
