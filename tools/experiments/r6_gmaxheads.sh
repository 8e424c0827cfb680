#!/bin/bash
# round 6: whole-grid max pool + heads in one launch for per-pose calls -- bits, seam latency with / without
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_concurrency.py tests/test_gpu_dense_split.py tests/test_gpu_custom_model.py tests/test_host_adapter.py -m gpu -x -q 2>&1 | tail -4
python - <<'PY'
import os, numpy as np
from gnina_amd import capi
capi.init(0)
G = np.load("tests/golden/cnn_goldens.npz")
for names in (["dense"], ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]):
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{names[0]}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    s = capi.Scorer(names); s.set_receptor(rec_xyz, rec_smt)
    a = [s.score_batch(poses[k:k + 1], lig_smt) for k in range(4)]
    with capi.option("MI_GNINA_NO_GMAX_FUSE", 1):
        b = [s.score_batch(poses[k:k + 1], lig_smt) for k in range(4)]
    big = s.score_batch(np.concatenate([poses] * 4), lig_smt)
    ok = all(np.array_equal(x[q], y[q]) for x, y in zip(a, b) for q in ("pose", "affinity", "loss"))
    ok2 = all(a[k]["pose"][0] == big["pose"][k] and a[k]["affinity"][0] == big["affinity"][k] for k in range(4))
    print(names[0], "fused == separate:", ok, " B=1 == batch of 16:", ok2)
PY
timeout 300 python tools/experiments/seam_b1_ensemble.py
timeout 300 python tools/experiments/seam_b1_ensemble.py MI_GNINA_NO_GMAX_FUSE=1
