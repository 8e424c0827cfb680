"""B = 1 latency of gnina's default ensemble: lanes enqueued round robin (MI_GNINA_LANE_SLICE = 1, 2, 4) against one program
after the other (1000) and against one stream (MI_GNINA_NO_LANES=1); synthetic receptor of bench.py and the golden complex."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi, synth  # noqa: E402

capi.init(0)
names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
cases = {"golden": tuple(G[f"dense_1_3/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))}
rng = np.random.RandomState(0)
m0 = capi.Model("crossdock_default2018")
rx, rs = synth.make_receptor(rng, 2500, synth.mapped_types(m0.chan_of_smt(False)))
lx, ls = synth.make_ligand(rng, 32, synth.mapped_types(m0.chan_of_smt(True)))
cases["synthetic"] = (rx, rs, ls, synth.make_poses(rng, lx, 4))


def med(s, poses, lig_smt, n=200):
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        s.score_batch(poses[:1], lig_smt)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        s.score_batch(poses[:1], lig_smt)
        ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6


for tag, (rec_xyz, rec_smt, lig_smt, poses) in cases.items():
    s = capi.Scorer(names)
    s.set_receptor(rec_xyz, rec_smt)
    row = {}
    for rep in range(2):
        for sl in ("1", "2", "4", "1000"):
            with capi.option("MI_GNINA_LANE_SLICE", sl):
                row.setdefault("slice " + sl, []).append(round(med(s, poses, lig_smt)))
        with capi.option("MI_GNINA_NO_LANES", "1"):
            row.setdefault("one stream", []).append(round(med(s, poses, lig_smt)))
    print(tag, row)
