import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gnina_amd import capi, synth
capi.init(0)
import sys as _s
for name in (_s.argv[1:] or ["default2017"]):
    m = capi.Model(name); s = capi.Scorer([m])
    rng = np.random.RandomState(0)
    rec_xyz, rec_smt = synth.make_receptor(rng, 2500, synth.mapped_types(m.chan_of_smt(False)))
    lx, ls = synth.make_ligand(rng, 32, synth.mapped_types(m.chan_of_smt(True)))
    s.set_receptor(rec_xyz, rec_smt)
    poses = synth.make_poses(rng, lx, 1)
    for _ in range(20): s.score_batch(poses, ls)
    t=[]
    for _ in range(200):
        t0=time.perf_counter(); s.score_batch(poses, ls); t.append(time.perf_counter()-t0)
    print("wall us median", np.median(t)*1e6, "min", np.min(t)*1e6)
    s.enable_profile(True); s.score_batch(poses, ls); prof=s.profile(); s.enable_profile(False)
    rows = prof if isinstance(prof, list) else prof.get("kernels", prof)
    print([(r["kernel"], round(r["ms_total"]*1e3,1)) for r in rows], "sum us", sum(r["ms_total"] for r in rows)*1e3)
