#!/usr/bin/env python3
"""Workload for rocprofv3: the BFGS kernel at 8,192 chains (throughput regime) and at 64 chains (latency regime)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi, synth
from gnina_amd import vina_scene
from oracle import vina as V
capi.init(0)
sc = vina_scene.build(0); lig = sc["lig"]
gd = V.setup_grid_dims(sc["center"], sc["size"])
types = sorted(set(int(t) for t in lig["smt"] if t > 1))
vina = capi.Vina(); vina.set_receptor(sc["rec_xyz"], sc["rec_smt"])
vina.build_cache(list(gd.begin), list(gd.end), list(gd.n), types, 1e3); vina.set_ligand(lig)
rng = np.random.RandomState(1)
base = np.stack([synth.random_conf(rng, lig, sc["center"], 1.0) for _ in range(2048)])
for B in (8192, 64):
    confs = np.tile(base, (B // len(base) + 1, 1))[:B]
    for _ in range(2):
        vina.bfgs_batch(confs, (10.0, 10.0, 10.0))
