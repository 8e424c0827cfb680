#!/bin/bash
# round 6: fc_heads_kernel with its operands requested eight iterations ahead -- tests, per-pose latency of the single models
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gradient.py tests/test_gpu_h2.py tests/test_host_adapter.py tests/test_gpu_custom_model.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import time, numpy as np
from gnina_amd import capi
capi.init(0)
G = np.load("tests/golden/cnn_goldens.npz")
for name in ("default2017", "crossdock_default2018", "dense"):
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    s = capi.Scorer([name]); s.set_receptor(rec_xyz, rec_smt)
    for _ in range(50): s.score_batch(poses[:1], lig_smt)
    t = []
    for k in range(400):
        t0 = time.perf_counter(); s.score_batch(poses[k % 4:k % 4 + 1], lig_smt); t.append(time.perf_counter() - t0)
    print(f"{name}: {np.median(t) * 1e6:.0f} us per B = 1 call")
PY
python tools/experiments/dense_throughput.py | tail -1
