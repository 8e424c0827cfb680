#!/bin/bash
# round 5, GPU call 58: per-group ligand description caches + lanes enqueued round robin: tests, seam numbers, timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_concurrency.py tests/test_host_adapter.py tests/test_gpu_gradient.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ensemble or ragged or chunking or error" 2>&1 | tail -3
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
from gnina_amd import capi, synth
capi.init(0)
d = bench.config_seam_b1(capi, synth)
print(json.dumps({k: d[k] for k in ('default2017', 'default_ensemble')}))
PY
bash tools/experiments/r5_run57.sh 2>&1 | grep -v "d16_kernel\|k1s\|h2_16_kernel" | tail -22
