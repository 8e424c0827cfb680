"""gnina's default ensemble at B = 2 ... 64 poses per call with and without lanes (MI_GNINA_LANES_MAX_B): median us per call.
python tools/experiments/lanes_max_b.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi, synth  # noqa: E402

capi.init(0)
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
rec_xyz, rec_smt, lig_smt, poses = (G[f"{names[0]}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
many = np.concatenate([poses, synth.make_poses(np.random.RandomState(3), poses[0] - poses[0].mean(0), 60)])
s = capi.Scorer(names)
s.set_receptor(rec_xyz, rec_smt)
for _ in range(40):
    s.score_batch(many[:1], lig_smt)
for B in (2, 4, 8, 9, 12, 16, 24, 32, 64):
    out = []
    for mx in (1, 64):
        with capi.option("MI_GNINA_LANES_MAX_B", mx):
            for _ in range(10):
                r = s.score_batch(many[:B], lig_smt)
            t = []
            for _ in range(60):
                t0 = time.perf_counter()
                r = s.score_batch(many[:B], lig_smt)
                t.append(time.perf_counter() - t0)
            out.append((np.median(t) * 1e6, r["pose"].copy(), r["affinity"].copy()))
    same = np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    print(f"B = {B:2d}: one stream {out[0][0]:7.0f} us, lanes {out[1][0]:7.0f} us per call; same bits: {same}")
