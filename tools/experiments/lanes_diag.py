"""Which part of a small ensemble call deviates from the goldens: per-model scores at B = 1 against the reference's TorchScript
outputs, then the default ensemble with its models on their own streams (lanes) and serially (MI_GNINA_NO_LANES=1), several
times each (a race would show as run-to-run differences)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
base = names[0]
rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
print("poses", len(poses))
for n in names:
    s = capi.Scorer([n])
    s.set_receptor(rec_xyz, rec_smt)
    one = [s.score_batch(poses[b:b + 1], lig_smt) for b in range(len(poses))]
    allb = s.score_batch(poses, lig_smt)
    dp = max(abs(float(one[b]["pose"][0]) - float(G[n + "/pose"][b])) for b in range(len(poses)))
    da = max(abs(float(one[b]["affinity"][0]) - float(G[n + "/affinity"][b])) for b in range(len(poses)))
    same = all(one[b]["pose"][0] == allb["pose"][b] and one[b]["affinity"][0] == allb["affinity"][b] for b in range(len(poses)))
    print(f"{n}: B=1 vs golden dpose {dp:.2e} daff {da:.2e}; B=1 bits == batch bits: {same}; affinity {[float(o['affinity'][0]) for o in one]}")
pose_ref = np.mean([G[n + "/pose"] for n in names], axis=0)
aff_ref = np.mean([G[n + "/affinity"] for n in names], axis=0)
for mode in ("lanes", "serial"):
    with capi.option("MI_GNINA_NO_LANES", None if mode == "lanes" else "1"):
        s = capi.Scorer(names)
        s.set_receptor(rec_xyz, rec_smt)
        runs = []
        for rep in range(6):
            one = [s.score_batch(poses[b:b + 1], lig_smt) for b in range(len(poses))]
            runs.append(np.array([[float(o["pose"][0]), float(o["affinity"][0])] for o in one]))
        allb = s.score_batch(poses, lig_smt)
        r0 = runs[0]
        print(f"{mode}: daff vs golden mean {np.abs(r0[:, 1] - aff_ref).max():.2e} dpose {np.abs(r0[:, 0] - pose_ref).max():.2e}; "
              f"repeatable {all(np.array_equal(r, r0) for r in runs)}; spread {max(np.abs(r - r0).max() for r in runs):.2e}; "
              f"batch daff {np.abs(allb['affinity'] - aff_ref).max():.2e} batch==single {np.array_equal(allb['affinity'], r0[:, 1].astype(np.float32))}")
        print("   affinity", r0[:, 1], "want", aff_ref)
