"""B = 1 gradient calls (score + d loss / d atoms: CNN refinement's evaluation, torch_model.cpp:197-221) of one model or of
gnina's default ensemble, for a kernel trace: python tools/experiments/b1_grad_timeline.py [default2017 | ensemble] [NAME=V ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
which = sys.argv[1] if len(sys.argv) > 1 else "default2017"
for o in sys.argv[2:]:
    capi.set_option(*o.split("=", 1))
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"] if which == "ensemble" else [which]
rec_xyz, rec_smt, lig_smt, poses = (G[f"{names[0]}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
s = capi.Scorer(names)
s.set_receptor(rec_xyz, rec_smt)
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    s.score_grad(poses[:1], lig_smt)
ts = []
for rep in range(40):
    t0 = time.perf_counter()
    s.score_grad(poses[:1], lig_smt)
    ts.append(time.perf_counter() - t0)
print("median call %.0f us" % (np.median(ts) * 1e6))
