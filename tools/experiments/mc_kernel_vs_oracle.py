"""debug: kernel MC vs oracle MC (mt19937), shortest possible chains"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gnina_amd import capi
from oracle import vina as V
from gnina_amd import vina_scene
capi.init(0)
sc = vina_scene.build(0)
lig = sc["lig"]
gd = V.setup_grid_dims(sc["center"], sc["size"])
T = V.Tables()
types = sorted(set(int(t) for t in lig["smt"] if t > 1))
grids = {t: V.cache_populate(T, gd, sc["rec_xyz"], sc["rec_smt"], t) for t in types}
S = V.Scene(T, gd, grids, V.LigandHandle(lig))
v = capi.Vina()
v.set_receptor(sc["rec_xyz"], sc["rec_smt"])
v.build_cache(list(gd.begin), list(gd.end), list(gd.n), types, 1e3)
v.set_ligand(lig)
c1, c2 = list(gd.begin), list(gd.end)
np.set_printoptions(precision=5, suppress=True, linewidth=200)
for B in (2, 600):
    seeds = np.arange(100, 100 + B, dtype=np.uint64)
    for iters in (0, 1, 2):
        n, e, cf, xyz, ev = v.mc_batch(seeds, c1, c2, capi.McParams.default(1, iters, 8))
        for b in range(2):
            e0, cf0, xyz0, ev0 = V.mc_chain(S, c1, c2, int(seeds[b]), 1, iters, 8)
            print("B", B, "iters", iters, "seed", seeds[b], "kernel e", e[b, 0], "oracle e", e0[0], "evals", ev[b], ev0)
            print("  k", cf[b, 0]); print("  o", cf0[0])
