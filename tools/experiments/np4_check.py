import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gnina_amd import capi, synth
capi.init(0)
for name in ("default2017", "dense"):
    m = capi.Model(name); s = capi.Scorer([m])
    rng = np.random.RandomState(0)
    rec_xyz, rec_smt = synth.make_receptor(rng, 2500, synth.mapped_types(m.chan_of_smt(False)))
    lx, ls = synth.make_ligand(rng, 32, synth.mapped_types(m.chan_of_smt(True)))
    s.set_receptor(rec_xyz, rec_smt)
    for B in (1024, 67):
        poses = synth.make_poses(rng, lx, B)
        capi.set_option("MI_GNINA_H2_WLDS", "2"); ref = s.score_batch(poses, ls)
        capi.set_option("MI_GNINA_H2_WLDS", "4"); got = s.score_batch(poses, ls)
        print(name, B, "NP=4 same bits:", np.array_equal(got["pose"], ref["pose"]) and np.array_equal(got["affinity"], ref["affinity"]), flush=True)
    poses = synth.make_poses(rng, lx, 1024)
    for w in ("2", "4"):
        capi.set_option("MI_GNINA_H2_WLDS", w)
        for _ in range(30): s.score_batch(poses, ls)
        s.enable_profile(True); s.score_batch(poses, ls); prof = s.profile(); s.enable_profile(False)
        rows = prof if isinstance(prof, list) else prof.get("kernels", prof)
        c = [r for r in rows if "conv3_s24" in r["kernel"]][:1]
        t0 = time.perf_counter()
        for _ in range(30): s.score_batch(poses, ls)
        dt = (time.perf_counter() - t0) / 30
        print(name, "WLDS", w, f"{1024/dt:.0f} poses/s, first conv {c[0]['ms_total']:.3f} ms", flush=True)
