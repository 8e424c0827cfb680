#!/usr/bin/env python3
"""Per-evaluation latency of the Vina wave kernel by mode + BFGS per-eval time (A/B experiments)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi, synth
from gnina_amd import vina_scene
from oracle import vina as V
import time
capi.init(0)
sc = vina_scene.build(0); lig = sc["lig"]
gd = V.setup_grid_dims(sc["center"], sc["size"])
types = sorted(set(int(t) for t in lig["smt"] if t > 1))
vina = capi.Vina(); vina.set_receptor(sc["rec_xyz"], sc["rec_smt"])
vina.build_cache(list(gd.begin), list(gd.end), list(gd.n), types, 1e3); vina.set_ligand(lig)
rng = np.random.RandomState(1)
c64 = np.stack([synth.random_conf(rng, lig, sc["center"], 1.0) for _ in range(64)])
out = {name: round(vina.eval_latency_us(c64, mode), 2) for name, mode in
       (("coords", 3), ("grid", 2), ("pairs", 4), ("energy", 0), ("grad", 1))}
v = (10.0, 10.0, 10.0)
vina.bfgs_batch(c64, v)
t0 = time.perf_counter(); e, cf, g, ev = vina.bfgs_batch(c64, v); dt = time.perf_counter() - t0
out["bfgs_us_per_eval"] = round(1e6 * dt / ev.max(), 2)
out["bfgs_e_sum"] = float(e.sum())
print(os.environ.get("MI_GNINA_LIB", "default").split("/")[-1], json.dumps(out))
