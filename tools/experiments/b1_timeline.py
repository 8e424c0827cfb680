"""B = 1 calls of gnina's default ensemble (run under rocprofv3 --kernel-trace: tools/experiments/r5_calls.sh 57)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
for o in sys.argv[1:]:
    capi.set_option(*o.split("=", 1))
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
rec_xyz, rec_smt, lig_smt, poses = (G[f"dense_1_3/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
s = capi.Scorer(["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"])
s.set_receptor(rec_xyz, rec_smt)
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    s.score_batch(poses[:1], lig_smt)
ts = []
for rep in range(40):
    t0 = time.perf_counter()
    s.score_batch(poses[:1], lig_smt)
    ts.append(time.perf_counter() - t0)
print("median call %.0f us" % (np.median(ts) * 1e6))
