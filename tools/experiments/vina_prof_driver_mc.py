#!/usr/bin/env python3
"""Workload for rocprofv3: the Monte-Carlo kernel with 64 chains (teams of 4 waves) and 4,096 chains (one wave each)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi
from gnina_amd import vina_scene
from oracle import vina as V
capi.init(0)
sc = vina_scene.build(0); lig = sc["lig"]
gd = V.setup_grid_dims(sc["center"], sc["size"])
types = sorted(set(int(t) for t in lig["smt"] if t > 1))
vina = capi.Vina(); vina.set_receptor(sc["rec_xyz"], sc["rec_smt"])
vina.build_cache(list(gd.begin), list(gd.end), list(gd.n), types, 1e3); vina.set_ligand(lig)
c1, c2 = list(gd.begin), list(gd.end)
for B, steps in ((64, 400), (4096, 100)):
    P = capi.McParams.default(steps, (25 + 32) // 3, 50)
    n, e, cf, xyz, ev = vina.mc_batch(np.arange(1, B + 1, dtype=np.uint64) * np.uint64(7919), c1, c2, P)
    print(B, steps, int(ev.sum()))
