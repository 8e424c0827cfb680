import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnina_amd import capi
from oracle import cnn_ref, cnn_refine, vina as ovina
from tests import vina_scene
capi.init(0)
sc = vina_scene.build(seed=3)
lig = sc["lig"]; lig["smt"] = lig["smt"].copy(); lig["smt"][[5, 20]] = 1
v = capi.Vina(); v.set_ligand(lig); olig = ovina.LigandHandle(lig)
name = "crossdock_default2018"
s = capi.Scorer([name]); s.set_receptor(sc["rec_xyz"], sc["rec_smt"]); v.set_receptor(sc["rec_xyz"], sc["rec_smt"])
gd = ovina.setup_grid_dims(sc["center"], sc["size"])
lo, hi = np.array(list(gd.begin), np.float32), np.array(list(gd.end), np.float32)
box = capi.CnnBox.make(23.5, lo, hi, slope=1e3)
W = os.path.join(os.path.dirname(capi.__file__), "weights")
blob = cnn_ref.Blob(os.path.join(W, name + ".mgw"))
seeds = np.array([11, 12, 13], dtype=np.uint64)
for steps, iters in ((1, 0), (1, 1), (2, 1), (2, 2)):
    P = capi.McParams.default(steps, iters, 10)
    n, e, cf, xyz, ev, cnn = v.mc_cnn_batch(s, seeds, lo, hi, P, box, level_all=True)
    for b, seed in enumerate(seeds):
        nc = cnn_refine.NonCacheCnn([blob], sc["rec_xyz"], sc["rec_smt"], olig, (lo, hi), 23.5)
        nc.slope = 1e3
        out, evals = cnn_refine.mc_cnnall(nc, int(seed), steps, lo, hi, iters, num_saved=10)
        print("steps", steps, "iters", iters, "seed", seed, "n", n[b], len(out), "evals", ev[b], evals, "e", e[b, :n[b]], [float(o[0]) for o in out])
        print("   dconf", np.abs(out[0][1] - cf[b, 0]).max(), np.round(cf[b, 0][:7], 4), np.round(out[0][1][:7], 4))
