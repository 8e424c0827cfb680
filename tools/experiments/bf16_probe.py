import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnina_amd import capi, synth
capi.init(0)
G = np.load("tests/golden/cnn_goldens.npz")
for name in []:
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    s = capi.Scorer([name]); s.set_receptor(rec_xyz, rec_smt)
    a = s.score_batch(poses, lig_smt)
    s.set_precision(True)
    b = s.score_batch(poses, lig_smt)
    print(name, "pose", a["pose"], b["pose"], "dpose", np.abs(a["pose"]-b["pose"]).max(), "daff", np.abs(a["affinity"]-b["affinity"]).max(), flush=True)
rng = np.random.RandomState(0)
m0 = capi.Model("dense_1_3")
rt, lt = synth.mapped_types(m0.chan_of_smt(False)), synth.mapped_types(m0.chan_of_smt(True))
rec_xyz, rec_smt = synth.make_receptor(rng, 2500, rt)
lx, ls = synth.make_ligand(rng, 32, lt)
for label, model, B in (("dense 48", "dense", 1024), ("default2017", "default2017", 1024), ("dense_1_3 96", capi.Model("dense_1_3", resolution=0.25, dimension=23.75), 256)):
    s = capi.Scorer([model]); s.set_receptor(rec_xyz, rec_smt)
    poses = synth.make_poses(rng, lx, B)
    for bf in (False, True):
        s.set_precision(bf)
        o = s.score_batch(poses, ls)
        t0 = time.perf_counter()
        for _ in range(3): o2 = s.score_batch(poses, ls)
        dt = (time.perf_counter() - t0) / 3
        if not bf: ref = o
        print(label, "bf16" if bf else "fp32", "poses/s", B / dt, "dpose", np.abs(o["pose"]-ref["pose"]).max(), "daff", np.abs(o["affinity"]-ref["affinity"]).max(), "aff rms", float(np.sqrt(((o["affinity"]-ref["affinity"])**2).mean())), flush=True)
    if os.environ.get("PROF") and ("96" in label or label == "dense 48"):
        s.enable_profile(True); s.score_batch(poses, ls)
        for k in s.profile(): print("   ", k["kernel"], round(k["ms_total"],3), "ms", round(k["flops"]/k["ms_total"]/1e9,1), "TF")
