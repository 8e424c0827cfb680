"""Which model of a small ensemble call deviates when its models run on their own streams (lanes), and what it takes:
per-model outputs (mi_scorer_last_model_outputs) of 40 B = 1 calls per setting against the serial run's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
base = "dense_1_3"
rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))


def run(names, lanes, reps=40, reset_receptor=False, two=False):
    with capi.option("MI_GNINA_NO_LANES", None if lanes else "1"):
        s = capi.Scorer(names)
        s.set_receptor(rec_xyz, rec_smt)
        s2 = None
        if two:
            s2 = capi.Scorer(names)
            s2.set_receptor(rec_xyz, rec_smt)
        out = []
        for rep in range(reps):
            b = rep % len(poses)
            if reset_receptor:
                s.set_receptor(rec_xyz, rec_smt)
            s.score_batch(poses[b:b + 1], lig_smt)
            out.append([[float(x[0]) for x in s.last_model_outputs(m, 1)[:2]] for m in range(len(names))])
            if s2 is not None:
                s2.score_batch(poses[b:b + 1], lig_smt)
        return np.array(out)  # [rep][model][pose, aff]


for names in (["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"], ["crossdock_default2018_KD_4", "default2017"],
              ["dense_1_3", "dense_1_3_PT_KD_3"], ["default2017", "default2017"], ["crossdock_default2018", "crossdock_default2018_KD_4"]):
    ref = run(names, False)
    for kw in ({}, {"reset_receptor": True}, {"two": True}):
        got = run(names, True, **kw)
        d = np.abs(got - ref)
        bad = [(int(r), int(m), int(k), float(d[r, m, k])) for r, m, k in zip(*np.nonzero(d))]
        print(names, kw, "deviating (call, model, pose|aff, |d|):", len(bad), bad[:8])
