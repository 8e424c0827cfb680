import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnina_amd import capi, synth
capi.init(0)
rng = np.random.RandomState(0)
m0 = capi.Model("crossdock_default2018")
rt, lt = synth.mapped_types(m0.chan_of_smt(False)), synth.mapped_types(m0.chan_of_smt(True))
rec_xyz, rec_smt = synth.make_receptor(rng, 2500, rt)
lx, ls = synth.make_ligand(rng, 32, lt)
for models in (["default2017"], ["dense_1_3"]):
    s = capi.Scorer(models); s.set_receptor(rec_xyz, rec_smt)
    poses = synth.make_poses(rng, lx, 1)
    for _ in range(5): s.score_batch(poses, ls)
    t0 = time.perf_counter()
    for _ in range(50): s.score_batch(poses, ls)
    print(models, "B=1 wall us", (time.perf_counter()-t0)/50*1e6)
    s.enable_profile(True)
    s.score_batch(poses, ls)
    tot = 0
    for k in s.profile():
        print("   ", k["kernel"], round(k["ms_total"]*1e3, 1), "us"); tot += k["ms_total"]*1e3
    print("    sum of kernels", round(tot, 1), "us")
