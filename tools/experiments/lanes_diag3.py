"""Lanes, second version (every voxel group voxelized before the first lane starts): per-model outputs of B = 1 calls of
gnina's default ensemble with MI_GNINA_LANES=1 against the serial run's, and the call's wall time either way."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
base = "dense_1_3"
rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def run(names, lanes):
    with capi.option("MI_GNINA_LANES", "1" if lanes else None):
        s = capi.Scorer(names)
        s.set_receptor(rec_xyz, rec_smt)
        out, ts = [], []
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            s.score_batch(poses[:1], lig_smt)
        for rep in range(reps):
            b = rep % len(poses)
            t0 = time.perf_counter()
            s.score_batch(poses[b:b + 1], lig_smt)
            ts.append(time.perf_counter() - t0)
            out.append([[float(x[0]) for x in s.last_model_outputs(m, 1)[:2]] for m in range(len(names))])
        return np.array(out), float(np.median(ts)) * 1e6


for names in (["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"], ["crossdock_default2018_KD_4", "default2017"],
              ["dense_1_3", "dense_1_3_PT_KD_3"]):
    ref, t_ref = run(names, False)
    got, t_got = run(names, True)
    d = np.abs(got - ref)
    bad = [(int(r), int(m), int(k), float(d[r, m, k])) for r, m, k in zip(*np.nonzero(d))]
    print(names, "serial %.0f us, lanes %.0f us per call; deviating (call, model, pose|aff, |d|) of %d calls: %d" % (t_ref, t_got, reps, len(bad)), bad[:6])
