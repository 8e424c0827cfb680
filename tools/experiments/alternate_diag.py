"""No concurrency at all: two scorers on ONE host thread, called alternately (every call synchronous).  Does the second
scorer's presence change the first one's results?  (It would if results depended on which XCD a workgroup lands on.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
capi.set_option("MI_GNINA_NO_LANES", "1")
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
base = "dense_1_3"
rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
N = int(os.environ.get("DIAG_CALLS", "200"))
for vname, aname in (("crossdock_default2018_KD_4", "dense_1_3"), ("dense_1_3_PT_KD_3", "dense_1_3"), ("default2017", "dense_1_3")):
    v = capi.Scorer([vname]); v.set_receptor(rec_xyz, rec_smt)
    a = capi.Scorer([aname]); a.set_receptor(rec_xyz, rec_smt)
    ref = []
    for p in range(len(poses)):
        r = v.score_batch(poses[p:p + 1], lig_smt)
        ref.append((float(r["pose"][0]), float(r["affinity"][0])))
    bad = 0; worst = 0.0
    for rep in range(N):
        p = rep % len(poses)
        a.score_batch(poses[(rep * 7) % len(poses):(rep * 7) % len(poses) + 1], lig_smt)
        r = v.score_batch(poses[p:p + 1], lig_smt)
        d = max(abs(float(r["pose"][0]) - ref[p][0]), abs(float(r["affinity"][0]) - ref[p][1]))
        bad += d > 0; worst = max(worst, d)
    print(vname, "next to", aname, "(alternating, one thread): deviating calls", bad, "of", N, "max |d|", worst)
