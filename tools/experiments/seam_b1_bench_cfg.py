"""bench.py's seam_b1 configuration of gnina's default ensemble (synthetic 2,500-atom receptor, 32-atom ligand), B = 1:
median call time under the switches given on the command line (NAME=value ...)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi, synth  # noqa: E402

capi.init(0)
for o in sys.argv[1:]:
    capi.set_option(*o.split("=", 1))
rng = np.random.RandomState(0)
m0 = capi.Model("crossdock_default2018")
rt, lt = synth.mapped_types(m0.chan_of_smt(False)), synth.mapped_types(m0.chan_of_smt(True))
rec_xyz, rec_smt = synth.make_receptor(rng, 2500, rt)
lx, ls = synth.make_ligand(rng, 32, lt)
pose1 = synth.make_poses(rng, lx, 1)
for models in (["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"], ["dense_1_3"], ["crossdock_default2018_KD_4"]):
    s = capi.Scorer(models)
    s.set_receptor(rec_xyz, rec_smt)
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        s.score_batch(pose1, ls)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        s.score_batch(pose1, ls)
        ts.append(time.perf_counter() - t0)
    print(sys.argv[1:], models, f"median {np.median(ts) * 1e6:.0f} us, min {np.min(ts) * 1e6:.0f}", flush=True)
    del s
