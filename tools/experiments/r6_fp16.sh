#!/bin/bash
# round 6: MI_PRECISION_FP16 (h * h only) -- tolerance test, C5 forward rates per precision; ws bits test; concurrency tests; threads
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_h2.py tests/test_gpu_concurrency.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -30
timeout 600 python - <<'PY'
import json, sys
sys.argv=["bench.py"]
import bench
from gnina_amd import capi, synth
capi.init(0)
r = bench.config_c5(capi, synth)
print(json.dumps({k: (v if k in ("workload",) else {kk: vv for kk, vv in v.items() if kk.startswith("poses")} if isinstance(v, dict) and "poses_per_s_forward" in v else None) for k, v in r.items()}))
PY
timeout 300 python tools/experiments/seam_b1_ensemble.py
