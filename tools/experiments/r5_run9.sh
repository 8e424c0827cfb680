#!/bin/bash
# round 5, GPU call 9: how often does a small ensemble call deviate, and under what: 16 runs of the DLScorer driver per setting
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python - <<'PY'
import os, struct, subprocess, numpy as np
G = np.load("tests/golden/cnn_goldens.npz")
names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
base = names[0]
rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
with open("/tmp/atoms.bin", "wb") as f:
    f.write(struct.pack("<4i", len(rec_smt), len(lig_smt), len(poses), 0))
    f.write(rec_xyz.astype("<f4").tobytes()); f.write(rec_smt.astype("<i4").tobytes()); f.write(lig_smt.astype("<i4").tobytes()); f.write(poses.astype("<f4").tobytes())
ref = None
for env in ({"MI_GNINA_NO_LANES": "1"}, {}, {"AMD_SERIALIZE_KERNEL": "3"}, {"MI_GNINA_NO_DENSE_SPLIT": "1"}, {"MI_GNINA_LANES_MAX_B": "1"}, {"MI_GNINA_CONV_PATH": "0"}, {"MI_GNINA_NO_LANES": "1"}):
    e = dict(os.environ); e.update(env)
    bad = 0; detail = []
    for rep in range(16):
        r = subprocess.run(["gnina_amd/lib/test_host_scorer", "/tmp/atoms.bin", "gnina_amd/weights"], capture_output=True, text=True, env=e)
        vals = []
        for l in r.stdout.split("\n"):
            t = l.split()
            if l.startswith("single"): vals += [float(t[2]), float(t[3]), float(t[7])]
            if l.startswith("batch"): vals += [float(t[2]), float(t[3])]
        vals = np.array(vals)
        if ref is None: ref = vals
        key = "ref" if env.get("MI_GNINA_NO_DENSE_SPLIT") or env.get("MI_GNINA_CONV_PATH") else None
        base_ = ref
        if key:  # different arithmetic: compare with this setting's own first run
            if rep == 0: own = vals
            base_ = own
        d = np.abs(vals - base_)
        if d.max() > 0: bad += 1; detail.append((rep, int(d.argmax()), float(d.max())))
    print(env, "runs that differ from the reference run:", bad, "of 16", detail[:6])
PY
